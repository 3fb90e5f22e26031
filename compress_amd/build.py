"""In-tree build of libkcgpu.so (HIP kernels + C ABI) for gfx950.

    python -m compress_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects are cached under compress_amd/_build/.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# KC_BUILD_TAG=<tag> (measurement builds only: tools/ A/B scripts): objects under _build_<tag>/, library libkcgpu_<tag>.so, loaded
# when KC_LIB_TAG=<tag> is set (compress_amd/_lib.py); the product is the untagged library
TAG = os.environ.get("KC_BUILD_TAG", "")
OUT = os.path.join(HERE, "libkcgpu%s.so" % ("_" + TAG if TAG else ""))
BDIR = os.path.join(HERE, "_build" + ("_" + TAG if TAG else ""))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-munsafe-fp-atomics"] + os.environ.get("KC_EXTRA_FLAGS", "").split()
# KC_FILE_FLAGS="kc_zstd_entropy.hip=-Os;kc_s2.hip=-O2 -DX=1" (measurement builds): flags appended for single source files
FILE_FLAGS = dict((kv.split("=", 1)[0].strip(), kv.split("=", 1)[1].split()) for kv in os.environ.get("KC_FILE_FLAGS", "").split(";") if "=" in kv)


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    os.makedirs(BDIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "kcgpu.h"))
    objs = []
    procs = []
    for s in srcs:
        sp = os.path.join(CSRC, s)
        op = os.path.join(BDIR, s + ".o")
        objs.append(op)
        if force or _newer(sp, op) or any(_newer(h, op) for h in hdrs):
            cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(s, []) + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("== %s failed ==\n%s\n" % (s, out.decode(errors="replace")))
        elif verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or not os.path.exists(OUT):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
