"""Sizing builds of the NT GEMM (csrc/tcn.hip gemm_nt_lds_kernel, -D switches): main loop vs launch + first stage + epilogue.
  python tools/diag/nt_variants.py --build     (CPU)
  python tools/diag/nt_variants.py             (GPU box: tools/diag/bench_nt.py once per variant)"""
import glob, os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
VARIANTS = {"base": [], "noloop": ["-DNT_DIAG_NOLOOP"], "noepi": ["-DNT_DIAG_NOEPI"], "noloop_noepi": ["-DNT_DIAG_NOLOOP", "-DNT_DIAG_NOEPI"]}
so = lambda n: os.path.join(root, "tools", "diag", f"libnt_{n}.so")
if "--build" in sys.argv:
    csrc = os.path.join(root, "generative-audio_amd", "csrc")
    objs = [os.path.join(root, "generative-audio_amd", "build", os.path.basename(f)[:-4] + ".o") for f in sorted(glob.glob(csrc + "/*.hip"))
            if not f.endswith("tcn.hip")]
    for n, flags in VARIANTS.items():
        o = f"/tmp/tcn_nt_{n}.o"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"),
                               "-I" + csrc, "-Wno-unused-value", *flags, "-c", os.path.join(csrc, "tcn.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so(n), o] + objs)
        print("built", so(n), flush=True)
    sys.exit(0)
for n in VARIANTS:
    print("==", n, flush=True)
    subprocess.call([sys.executable, os.path.join(root, "tools", "diag", "bench_nt.py")], env=dict(os.environ, NPPC_HIP_LIB=so(n)))
