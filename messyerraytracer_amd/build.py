"""Builds libmrt_hip.so (the C-ABI library: HIP kernels for gfx950 + host-side
scene preparation) in-tree with hipcc.  hipcc cross-compiles without a GPU."""
import concurrent.futures
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmrt_hip.so")
OBJ = os.path.join(HERE, "_obj")  # object files of the last build (git-ignored; not needed at run time)
HOST_TEST = os.path.join(HERE, "host_mirror_test")
HOST_CPU_TEST = os.path.join(HERE, "host_cpu_test")
HOST_TLAS_TEST = os.path.join(HERE, "host_tlas_test")

SOURCES = ["kernels.hip", "api.hip", "group.hip", "device_build.hip", "host/scene_prep.cpp", "host/bvh_builder.cpp",
           "host/two_level_prep.cpp"]
HEADERS = ["mrt_internal.h", "packet_kernel.h", "packet_asm_kernel.h", "packet_rows_kernel.h", "packet_quad_kernel.h", "two_level_kernel.h", "lane_persistent_kernel.h", "../../include/mrt_hip.h", "host/gpu_ray_caster.hpp", "host/ray_dispatcher.hpp",
           "host/host_types.hpp", "host/cpu_backend.hpp", "host/ray_tracer_server.hpp"]
# -Xarch_host -mfma: explicit fmaf() calls of the host code (the 8-wide collapse verifies every quantised
# box with the kernel's own fma) become one instruction instead of a libm call; nothing is contracted
# implicitly (-ffp-contract=off), so every result is unchanged.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-result", "-Xarch_host", "-mfma"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libmrt_hip.so cannot be built")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """MRT_WITH_QUAD=1 in the environment also compiles the four-wide packet walk (packet_quad_kernel.h: an experiment kept
    for the record, held to the oracle by the packet tests, slower than the default on every measured config)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if force or _stale(LIB, deps):
        extra = ["-DMRT_WITH_QUAD"] if os.environ.get("MRT_WITH_QUAD") == "1" else []
        extra += os.environ.get("MRT_EXTRA_DEFINES", "").split()  # A/B builds of the tools (e.g. -DMRT_ASM_KPF=0)
        # one object per translation unit, compiled side by side (kernels.hip alone is over a minute), then one link
        os.makedirs(OBJ, exist_ok=True)
        objs = [os.path.join(OBJ, s.replace("/", "_") + ".o") for s in SOURCES]

        def compile_one(job):
            src, obj = job
            cmd = [_hipcc()] + FLAGS + extra + ["-c", src, "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
            return r.stderr

        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
            logs = list(pool.map(compile_one, zip(srcs, objs)))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB, "-pthread"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
        if r.returncode != 0:
            raise RuntimeError("hipcc link failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose:
            print("".join(logs) + r.stderr)
    return LIB


def build_host_test(force: bool = False) -> str:
    """C++ test driver for the GPURayCaster / RayDispatcher mirrors (links the C-ABI)."""
    src = os.path.join(CSRC, "host", "host_mirror_test.cpp")
    deps = [src, LIB] + [os.path.join(CSRC, h) for h in HEADERS]
    if force or _stale(HOST_TEST, deps):
        cmd = [_hipcc(), "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", src, "-o", HOST_TEST,
               "-L" + HERE, "-lmrt_hip", "-Wl,-rpath," + HERE, "-pthread"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
        if r.returncode != 0:
            raise RuntimeError("host test build failed:\n" + r.stdout + r.stderr)
    return HOST_TEST


def build_host_cpu_test(force: bool = False) -> str:
    """C++ test driver for the RayTracerServer mirror over the router's CPU backend (links the C-ABI for the host-side builder)."""
    src = os.path.join(CSRC, "host", "host_cpu_test.cpp")
    deps = [src, LIB] + [os.path.join(CSRC, h) for h in HEADERS]
    if force or _stale(HOST_CPU_TEST, deps):
        cmd = [_hipcc(), "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", src, "-o", HOST_CPU_TEST,
               "-L" + HERE, "-lmrt_hip", "-Wl,-rpath," + HERE, "-pthread"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
        if r.returncode != 0:
            raise RuntimeError("host cpu test build failed:\n" + r.stdout + r.stderr)
    return HOST_CPU_TEST


def build_host_tlas_test(force: bool = False) -> str:
    """C++ test driver for the router with a TLAS set (two-level scenes on the CPU and the device backend)."""
    src = os.path.join(CSRC, "host", "host_tlas_test.cpp")
    deps = [src, LIB] + [os.path.join(CSRC, h) for h in HEADERS]
    if force or _stale(HOST_TLAS_TEST, deps):
        cmd = [_hipcc(), "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", src, "-o", HOST_TLAS_TEST,
               "-L" + HERE, "-lmrt_hip", "-Wl,-rpath," + HERE, "-pthread"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
        if r.returncode != 0:
            raise RuntimeError("host tlas test build failed:\n" + r.stdout + r.stderr)
    return HOST_TLAS_TEST


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
    print(build_host_test(force=True))
    print(build_host_cpu_test(force=True))
    print(build_host_tlas_test(force=True))
