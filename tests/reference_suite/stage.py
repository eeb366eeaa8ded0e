"""Stage the reference's OWN test files -- and its own benchmark script -- next to the drop-in, unmodified, so that they run
against it.

Source: /root/reference/tests/test_optimizer.py (3 optimizers x 7 robots x 2 hands, 100 random retargets each) and
/root/reference/tests/test_retargeting_config.py (config parsing, dict configs, dummy free joints).  The files are
COPIED BYTE FOR BYTE into tests/reference_suite/_ref/ -- a git-ignored directory (like built .so files it travels to the
GPU box with the working tree but never enters history: reference sources are not vendored into this repository) -- and
a sha256 manifest of what was staged is written beside them.  __graft_entry__.build() and this directory's conftest.py
both call stage() whenever /root/reference is present; on the GPU box (no /root/reference) the already staged files
are used as they are.

Round 6: the reference's measurement template, /root/reference/example/profiling/profile_online_retargeting.py
(7 robots x {vector, DexPilot}: 621 fixture frames, one SeqRetargeting.retarget per frame, prints 14 "fps" lines), and the pickle
it reads (human_joint_right.pkl) are staged the same way under _ref/example/profiling/; the script finds the URDFs at
<three levels up>/assets/robots/hands (profile_online_retargeting.py:40-43), i.e. _ref/assets -> the package's assets.
run_profile_script.py runs its main() with `dex_retargeting` aliased to the drop-in.

The reference tests locate their inputs relative to their own path (test_optimizer.py:20-22,
test_retargeting_config.py:36-38): <parent of tests dir>/assets/robots/hands, <...>/dex_retargeting/configs and
<...>/src/dex_retargeting/configs.  lay_out() creates those three paths under tests/reference_suite/ as symlinks to
the package's own assets/ and configs/.
"""
import hashlib
import json
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_TESTS = "/root/reference/tests"
STAGED = os.path.join(HERE, "_ref")
FILES = ("test_optimizer.py", "test_retargeting_config.py")
REF_PROFILING = "/root/reference/example/profiling"
PROFILE_DIR = os.path.join(STAGED, "example", "profiling")
PROFILE_FILES = ("profile_online_retargeting.py", "human_joint_right.pkl")


def stage() -> bool:
    """True when tests/reference_suite/_ref/ holds both files (freshly copied if the reference is here)."""
    if os.path.isdir(REF_TESTS):
        os.makedirs(STAGED, exist_ok=True)
        manifest = {}
        for name in FILES:
            src, dst = os.path.join(REF_TESTS, name), os.path.join(STAGED, name)
            with open(src, "rb") as f:
                data = f.read()
            if not os.path.exists(dst) or open(dst, "rb").read() != data:
                shutil.copyfile(src, dst)
            manifest[name] = {"sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data), "from": src}
        if os.path.isdir(REF_PROFILING):
            os.makedirs(PROFILE_DIR, exist_ok=True)
            for name in PROFILE_FILES:
                src, dst = os.path.join(REF_PROFILING, name), os.path.join(PROFILE_DIR, name)
                with open(src, "rb") as f:
                    data = f.read()
                if not os.path.exists(dst) or open(dst, "rb").read() != data:
                    shutil.copyfile(src, dst)
                manifest["example/profiling/" + name] = {"sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data), "from": src}
            _link(os.path.join(REPO, "dex_retargeting_amd", "assets"), os.path.join(STAGED, "assets"))
        with open(os.path.join(STAGED, "MANIFEST.json"), "w") as f:
            json.dump(manifest, f, indent=1)
    return all(os.path.exists(os.path.join(STAGED, n)) for n in FILES)


def profile_script_staged() -> bool:
    return all(os.path.exists(os.path.join(PROFILE_DIR, n)) for n in PROFILE_FILES) and os.path.exists(os.path.join(STAGED, "assets"))


def _link(target: str, link: str) -> None:
    os.makedirs(os.path.dirname(link), exist_ok=True)
    if os.path.islink(link):
        if os.path.realpath(link) == os.path.realpath(target):
            return
        os.unlink(link)
    elif os.path.exists(link):
        return
    os.symlink(os.path.relpath(target, os.path.dirname(link)), link)


def lay_out() -> None:
    pkg = os.path.join(REPO, "dex_retargeting_amd")
    _link(os.path.join(pkg, "assets"), os.path.join(HERE, "assets"))
    _link(os.path.join(pkg, "configs"), os.path.join(HERE, "dex_retargeting", "configs"))
    _link(os.path.join(pkg, "configs"), os.path.join(HERE, "src", "dex_retargeting", "configs"))


if __name__ == "__main__":
    lay_out()
    print("staged" if stage() else "reference tests not available")
