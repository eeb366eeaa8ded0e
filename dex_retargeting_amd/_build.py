"""Build libdexr.so (hipcc, gfx950) in-tree.  Used by __graft_entry__.build(); hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(REPO, "include")
BUILD = os.environ.get("DEXR_BUILD_DIR") or os.path.join(REPO, "build")
LIB = os.environ.get("DEXR_LIB_OUT") or os.path.join(HERE, "libdexr.so")
BUCKETS = (4, 8, 16, 24, 32)
CHAIN_BUCKETS = (4,)
BIG_BUCKETS = (16, 24, 32)
VARIANTS = ((0, 0), (1, 0), (1, 1), (1, 2))  # (float64?, mode): f32 solve, f64 solve, f64 eval, f64 fk
BIG_HEADER = os.path.join(CSRC, "dexr_big.hpp")
QUAD_HEADER = os.path.join(CSRC, "dexr_quad.hpp")
QUAD_BUCKETS = (16, 24)
HEADERS = [os.path.join(CSRC, "dexr_kernel.hpp"), os.path.join(CSRC, "dexr_launch.hpp"), os.path.join(CSRC, "dexr_tip.hpp"),
           os.path.join(CSRC, "dexr_math.hpp"),
           os.path.join(INCLUDE, "dexr.h"), os.path.join(INCLUDE, "dexr_tables.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}"] + \
    os.environ.get("DEXR_EXTRA_FLAGS", "").split()
# The SLP vectoriser turns the 3-vector arithmetic of the register kernels into v_pk_* pairs that it then has to
# assemble with v_mov_b32 (342 moves in the 4-joint chain kernel) and that need aligned register pairs: without it the
# same kernel has 9 % fewer VALU instructions, 126 instead of 156 VGPRs (4 waves per SIMD, no scratch) and runs
# 25 % faster (Allegro vector 0.108 -> 0.081 ms per 65 536 frames; Shadow vector 9.4 -> 4.3 ms).  The LDS kernel
# (dexr_big) measured no gain and keeps the default.
# float32 divisions / square roots of the solver (step scaling, Huber weights, Cholesky pivots) do not need IEEE
# rounding or denormal support -- parity is measured against the float64 oracle: 2.5-ulp v_rcp/v_rsq sequences and
# flushed denormals save another 8 % of the chain kernel's VALU instructions.
NO_SLP = ["-fno-slp-vectorize", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fgpu-flush-denormals-to-zero"]


# Which sources decide the code a bench workload's dominant kernel runs (used to key the committed rocprofv3 PMC
# summaries, profiles/pmc_<workload>.json: bench.py attaches their counters to a line only while this hash is unchanged).
_COMMON_SOURCES = ["dexr_api.hip", "dexr_launch.hpp", "../../include/dexr.h", "../../include/dexr_tables.h"]
KERNEL_SOURCES = {
    "allegro_vector": ["dexr_kernel.hpp", "dexr_tip.hpp", "dexr_math.hpp", "dexr_inst.hip"],
    "shadow_dexpilot": ["dexr_wide.hpp", "dexr_wide_inst.hip", "dexr_big.hpp", "dexr_math.hpp"],
    "leap_position": ["dexr_wide.hpp", "dexr_wide_inst.hip", "dexr_big.hpp", "dexr_math.hpp"],
    # sub-records of the default line (bench.py --probe): the float64 / cold-start launches of the headline config, the general
    # kernel; "mixed_fleet" is absent on purpose: four robots = every kernel family, i.e. all sources (the fallback)
    "allegro_vector_f64": ["dexr_kernel.hpp", "dexr_tip.hpp", "dexr_math.hpp", "dexr_inst.hip"],
    "allegro_vector_cold": ["dexr_kernel.hpp", "dexr_tip.hpp", "dexr_math.hpp", "dexr_inst.hip"],
    "general_kernel": ["dexr_gen.hpp", "dexr_gen_inst.hip", "dexr_kernel.hpp", "dexr_big.hpp", "dexr_math.hpp"],
}


def source_hash(workload: str) -> str:
    """sha256[:16] over the sources (names + contents) and compiler flags that produce `workload`'s dominant kernel."""
    import hashlib

    names = sorted(set(KERNEL_SOURCES.get(workload) or [f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]) | set(_COMMON_SOURCES))
    h = hashlib.sha256()
    for n in names:
        h.update(os.path.basename(n).encode())
        with open(os.path.normpath(os.path.join(CSRC, n)), "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS[:4] + NO_SLP).encode())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(BUILD, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    api_s, api_o = os.path.join(CSRC, "dexr_api.hip"), os.path.join(BUILD, "dexr_api.o")
    objs.append(api_o)
    if force or _stale(api_o, [api_s, os.path.join(CSRC, "dexr_hostctx.hpp"), os.path.join(CSRC, "dexr_gen.hpp")] + HEADERS):
        jobs.append((api_s, api_o, []))
    prep_s, prep_o = os.path.join(CSRC, "dexr_prep.hip"), os.path.join(BUILD, "dexr_prep.o")
    objs.append(prep_o)
    if force or _stale(prep_o, [prep_s, os.path.join(INCLUDE, "dexr.h")]):
        jobs.append((prep_s, prep_o, []))
    aux_s, aux_o = os.path.join(CSRC, "dexr_aux.hip"), os.path.join(BUILD, "dexr_aux.o")
    objs.append(aux_o)
    if force or _stale(aux_o, [aux_s, os.path.join(INCLUDE, "dexr.h")]):
        jobs.append((aux_s, aux_o, []))
    comm_s, comm_o = os.path.join(CSRC, "dexr_comm.hip"), os.path.join(BUILD, "dexr_comm.o")
    objs.append(comm_o)
    if force or _stale(comm_o, [comm_s, os.path.join(INCLUDE, "dexr.h")]):
        jobs.append((comm_s, comm_o, []))
    gen_s, gen_o = os.path.join(CSRC, "dexr_gen_inst.hip"), os.path.join(BUILD, "dexr_gen.o")
    objs.append(gen_o)
    if force or _stale(gen_o, [gen_s, os.path.join(CSRC, "dexr_gen.hpp")] + HEADERS):
        jobs.append((gen_s, gen_o, []))
    inst_s = os.path.join(CSRC, "dexr_inst.hip")
    # developer shortcut: DEXR_BUILD_ONLY="4,8" rebuilds only those buckets and reuses the other objects as they are
    # (only valid while KernelParams / the launcher signature are unchanged)
    only = os.environ.get("DEXR_BUILD_ONLY")
    only = None if not only else {int(v) for v in only.split(",")}
    # biggest kernels first so the thread pool stays busy
    for n in sorted(BUCKETS, reverse=True):
        for f64, mode in VARIANTS:
            if n == 32 and (f64, mode) == (0, 0):
                continue  # bucket 32 serves float32 requests with its float64 kernel (see dexr_launch.hpp)
            o = os.path.join(BUILD, f"dexr_inst_{n}_{f64}_{mode}.o")
            objs.append(o)
            if only is not None and n not in only and os.path.exists(o):
                continue
            if force or _stale(o, [inst_s] + HEADERS):
                jobs.append((inst_s, o, NO_SLP + [f"-DDEXR_NMAX={n}", f"-DDEXR_F64={f64}", f"-DDEXR_MODE={mode}"]))
    big_s = os.path.join(CSRC, "dexr_big_inst.hip")
    for n in BIG_BUCKETS:  # large-component kernel (Hessian in LDS, float64 kinematics)
        o = os.path.join(BUILD, f"dexr_big_{n}.o")
        objs.append(o)
        if only is not None and n not in only and os.path.exists(o):
            continue
        if force or _stale(o, [big_s, BIG_HEADER] + HEADERS):
            jobs.append((big_s, o, [f"-DDEXR_NMAX={n}"]))
    quad_s = os.path.join(CSRC, "dexr_quad_inst.hip")
    for n in QUAD_BUCKETS:  # four-lanes-per-frame kernel for dense components
        o = os.path.join(BUILD, f"dexr_quad_{n}.o")
        objs.append(o)
        if only is not None and n not in only and os.path.exists(o):
            continue
        if force or _stale(o, [quad_s, QUAD_HEADER, BIG_HEADER] + HEADERS):
            jobs.append((quad_s, o, NO_SLP + [f"-DDEXR_NMAX={n}"]))
    wide_s = os.path.join(CSRC, "dexr_wide_inst.hip")
    for n in (16, 24, 32):  # sixteen-lanes-per-frame kernel for dense components
        o = os.path.join(BUILD, f"dexr_wide_{n}.o")
        objs.append(o)
        if force or _stale(o, [wide_s, os.path.join(CSRC, "dexr_wide.hpp"), BIG_HEADER] + HEADERS):
            # the 16-row grid fits three waves per SIMD (168 VGPRs, 18 of them spilled; 11.8 KB of LDS per wave)
            jobs.append((wide_s, o, NO_SLP + [f"-DDEXR_NMAX={n}"] + (["-DDEXR_WIDE_MINW=3"] if n == 16 else [])))
    for n in (16, 24, 32):  # ... one frame per wave (SPRINT): the launch shape of small batches
        o = os.path.join(BUILD, f"dexr_wide_s_{n}.o")
        objs.append(o)
        if force or _stale(o, [wide_s, os.path.join(CSRC, "dexr_wide.hpp"), BIG_HEADER] + HEADERS):
            # (register budgets: two waves per SIMD for the 16- / 24-row grids (246 / 256 registers, 3 spilled at n = 24), one for
            # the 32-row grid (272): small batches do not need the occupancy)
            jobs.append((wide_s, o, NO_SLP + [f"-DDEXR_NMAX={n}", "-DDEXR_SPRINT=1", "-DDEXR_WIDE_MINW=1" if n == 32 else "-DDEXR_WIDE_MINW=2"]))
    for tag, defs in (("s_m", ["-DDEXR_SPRINT=1"]), ("s_mc", ["-DDEXR_SPRINT=1", "-DDEXR_MODCHOL=1"])):  # ... on the variable grid
        o = os.path.join(BUILD, f"dexr_wide_{tag}_16.o")
        objs.append(o)
        if force or _stale(o, [wide_s, os.path.join(CSRC, "dexr_wide.hpp"), BIG_HEADER] + HEADERS):
            jobs.append((wide_s, o, NO_SLP + ["-DDEXR_NMAX=16", "-DDEXR_MIMIC=1"] + defs))
    for tag, defs in (("m", []), ("mc", ["-DDEXR_MODCHOL=1"])):  # the same kernel on the grid of the optimised variables
        o = os.path.join(BUILD, f"dexr_wide_{tag}_16.o")         # (mimic joints), plain / modified Cholesky
        objs.append(o)
        if force or _stale(o, [wide_s, os.path.join(CSRC, "dexr_wide.hpp"), BIG_HEADER] + HEADERS):
            jobs.append((wide_s, o, NO_SLP + ["-DDEXR_NMAX=16", "-DDEXR_MIMIC=1"] + defs))
    red_s = os.path.join(CSRC, "dexr_red_inst.hip")
    for nvb in (8, 16):  # reduced-variable kernel (mimic models): Hessian of the variables in registers
        o = os.path.join(BUILD, f"dexr_red_{nvb}.o")
        objs.append(o)
        if force or _stale(o, [red_s, os.path.join(CSRC, "dexr_red.hpp"), BIG_HEADER] + HEADERS):
            jobs.append((red_s, o, NO_SLP + [f"-DDEXR_NV={nvb}"]))
    for n, f64 in ((4, 0), (8, 0), (4, 1), (8, 1)):  # small components with fleet / sequence addressing (EXT)
        o = os.path.join(BUILD, f"dexr_inst_ext_{n}_{f64}_0.o")
        objs.append(o)
        if force or _stale(o, [inst_s] + HEADERS):
            jobs.append((inst_s, o, NO_SLP + [f"-DDEXR_NMAX={n}", f"-DDEXR_F64={f64}", "-DDEXR_MODE=0", "-DDEXR_EXT=1"]))
    for n in CHAIN_BUCKETS:
        o = os.path.join(BUILD, f"dexr_inst_ext_chain_{n}_0_0.o")
        objs.append(o)
        if force or _stale(o, [inst_s] + HEADERS):
            jobs.append((inst_s, o, NO_SLP + [f"-DDEXR_NMAX={n}", "-DDEXR_F64=0", "-DDEXR_MODE=0", "-DDEXR_CHAIN=1", "-DDEXR_EXT=1"]))
    for f64 in (0, 1):  # tip pass of the serial-chain kernel (dexr_tip.hpp): float32, and float64 (the reference's arithmetic)
        for tag, defs in (("tip", []), ("ext_tip", ["-DDEXR_EXT=1"])):
            o = os.path.join(BUILD, f"dexr_inst_{tag}_4_{f64}_0.o")
            objs.append(o)
            if force or _stale(o, [inst_s] + HEADERS):
                jobs.append((inst_s, o, NO_SLP + ["-DDEXR_NMAX=4", f"-DDEXR_F64={f64}", "-DDEXR_MODE=0", "-DDEXR_CHAIN=1", "-DDEXR_TIP=1"] + defs))
    for n in CHAIN_BUCKETS:  # serial-chain specialisation, float32 solve only
        o = os.path.join(BUILD, f"dexr_inst_chain_{n}_0_0.o")
        objs.append(o)
        if force or _stale(o, [inst_s] + HEADERS):
            jobs.append((inst_s, o, NO_SLP + [f"-DDEXR_NMAX={n}", "-DDEXR_F64=0", "-DDEXR_MODE=0", "-DDEXR_CHAIN=1"]))

    def compile_one(job):
        s, o, defs = job
        cmd = [hipcc] + FLAGS + defs + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr[-4000:]}")
        if verbose:
            print("compiled", os.path.basename(o), file=sys.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_one, jobs))
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
