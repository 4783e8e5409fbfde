"""Build cactus_amd/lib/libcactus_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libcactus_hip.so")

SOURCES = ["quant_kernels.hip", "matmul_skinny.hip", "matmul_gemm.hip", "matmul_generic.hip",
           "matmul_f16.hip", "attention.hip", "attention_decode.hip", "attention_prefill.hip", "block_ops.hip", "conv_ops.hip", "sampler.hip", "comm.hip", "api.hip", "graph.cpp", "steps.cpp"]
HEADERS = ["common.hip.h", "block_math.hip.h", "kernels.h", "graph.h", os.path.join("..", "..", "include", "cactus_hip.h"), os.path.join("..", "..", "include", "cactus_hip_steps.h")]

# -ffp-contract=off: every fused multiply-add in the kernels is an explicit fmaf(), so the
# compiler must not introduce (or break) any -- results are compared bit for bit.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]
# Per-file additions.  The GEMM's fp32 rescale runs beside MFMAs, where a packed v_pk_fma_f32
# costs more than the two v_fma_f32 it replaces: keep the SLP vectorizer from packing them.
# The decode attention rescales its MFMA accumulators with VALU code: keep them in VGPRs (the default puts
# them in AGPRs and copies them back and forth around every use).
# The streaming matmuls read every MFMA result with VALU code right away (conversion, fp32 chain or LDS store): in
# VGPR form the batched INT8 kernel needs 80 registers instead of 92 + 4 (6 waves per SIMD instead of 5), the
# one-wave-per-tile kernel 126 instead of 168 + 32, and the copies out of the AGPRs go.
EXTRA_FLAGS = {"matmul_gemm.hip": ["-fno-slp-vectorize"],
               "attention_decode.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "matmul_skinny.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    mt = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > mt for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = ["hipcc"] + FLAGS + EXTRA_FLAGS.get(s, []) + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if force or procs or _stale(SO, objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    try:
        build_examples(force or bool(procs), verbose)
    except (subprocess.CalledProcessError, OSError) as e:    # the library itself is built: an example cannot block it
        print(f"warning: examples did not build: {e}", file=sys.stderr)
    return SO


def build_examples(force=False, verbose=False):
    """examples/*.cpp -> cactus_amd/lib/<name>: C++ host programs on the C ABI (host code only; they find the library
    next to themselves)."""
    root = os.path.dirname(HERE)
    inc = os.path.join(root, "include")
    hdrs = [os.path.join(inc, h) for h in ("cactus_hip.h", "cactus_hip_graph.h", "cactus_hip_steps.h")]
    built = []
    for name in ("qwen_graph_decode", "qwen_batched_decode", "qwen_prompt_decode", "host_pointer_calls"):
        src = os.path.join(root, "examples", name + ".cpp")
        exe = os.path.join(LIBDIR, name)
        if not os.path.exists(src):
            continue
        if force or _stale(exe, [src, SO] + hdrs):
            cmd = ["hipcc", "-O2", "-std=c++17", "-Wall", "-I", inc, src, "-L", LIBDIR, "-lcactus_hip", "-Wl,-rpath,$ORIGIN", "-o", exe]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        built.append(exe)
    return built


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
