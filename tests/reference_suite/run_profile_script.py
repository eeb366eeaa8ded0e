"""Run the reference's OWN benchmark, /root/reference/example/profiling/profile_online_retargeting.py:39-77, UNMODIFIED against
the drop-in: `dex_retargeting` (and every `dex_retargeting.<module>` it imports) is aliased to `dex_retargeting_amd` in
sys.modules, then the staged script (tests/reference_suite/_ref/example/profiling/, byte for byte the reference's, git-ignored)
is executed as `__main__` with runpy -- its own main(), its own timing loop (:18-36), its own print statements.

    python tests/reference_suite/run_profile_script.py            # prints the script's 15 lines
    python tests/reference_suite/run_profile_script.py --json     # + one JSON line: {"rows": [{"kind", "robot", "seconds", "fps"} x 14]}
    python tests/reference_suite/run_profile_script.py --json --passes 2   # main() twice in one process (second: warm HIP context)

bench.py runs this file in a process of its own (DETAIL line `reference_profile_script`); tests/reference_suite/
test_reference_profile_script.py runs it on the GPU and checks the 14 rows."""
import contextlib
import io
import json
import os
import re
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
if HERE not in sys.path:
    sys.path.insert(0, HERE)

ROW = re.compile(r"^(Vector|DexPilot) retargeting of (\S+) take ([0-9.eE+-]+)s in total, fps: ([0-9.eE+-]+)hz")


def alias_package() -> None:
    import importlib
    import pkgutil

    import dex_retargeting_amd as pkg

    sys.modules["dex_retargeting"] = pkg
    for m in pkgutil.iter_modules(pkg.__path__):
        if m.name.startswith("_") or m.ispkg or not os.path.exists(os.path.join(pkg.__path__[0], m.name + ".py")):
            continue
        sys.modules[f"dex_retargeting.{m.name}"] = importlib.import_module(f"dex_retargeting_amd.{m.name}")


def parse(text: str):
    rows = []
    for line in text.splitlines():
        m = ROW.match(line.strip())
        if m:
            rows.append({"kind": m.group(1).lower(), "robot": m.group(2), "seconds": float(m.group(3)), "fps": float(m.group(4))})
    return rows


def run():
    """-> (stdout text of the script's main(), parsed rows).  Raises when the script is not staged."""
    import stage

    stage.lay_out()
    stage.stage()
    if not stage.profile_script_staged():
        raise FileNotFoundError("tests/reference_suite/_ref/example/profiling/ is empty and /root/reference is not here: run "
                                "`python tests/reference_suite/stage.py` (or __graft_entry__.build()) where the reference is")
    alias_package()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        runpy.run_path(os.path.join(stage.PROFILE_DIR, "profile_online_retargeting.py"), run_name="__main__")
    text = buf.getvalue()
    return text, parse(text)


if __name__ == "__main__":
    passes = int(sys.argv[sys.argv.index("--passes") + 1]) if "--passes" in sys.argv else 1
    for p in range(passes):  # (pass 1 pays the process's first HIP context inside the script's first row; pass 2: a warm process)
        text, rows = run()
        sys.stdout.write(text)
        if "--json" in sys.argv:
            print("REFERENCE_PROFILE_SCRIPT " + json.dumps({"pass": p + 1, "rows": rows}))
        sys.stdout.flush()
