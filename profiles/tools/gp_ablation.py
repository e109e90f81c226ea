#!/usr/bin/env python3
"""Where the time of the GP factorisation kernels goes: builds variants of csrc/ste_gp.hip with one part compiled out each
(results are garbage, durations are not) and prints the build commands; profiles/r04_gp_ablation.txt is the result.

  python3 profiles/tools/gp_ablation.py            # writes scratch/gpx/ste_gp_x.hip + libste_<VARIANT>.so (needs lib/obj/*.o)
  then on the GPU box, per variant:  cd /tmp && STE_LIB_PATH=.../libste_<VARIANT>.so rocprofv3 --kernel-trace --stats \
        --output-format csv -d out -- python3 bench_gp.py --tracks 1000 --nobs 2000 --evals 2 --cpu-evals 0

Variants: BASE | NODIAG (no one-wave Cholesky / inverse of the diagonal block) | NOLEFT (no multiplication by the inverted
diagonal block: the accumulators are stored as they are) | NOINIT (accumulators start from zero instead of K[i][j]) | NOLOOP
(no panel loop at all: what is left is everything else).
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "ship-track-estimators_amd", "csrc", "ste_gp.hip")).read()
src = src.replace('#include "../../include/ste.h"', '#include "%s/include/ste.h"' % ROOT)


def rep(a, b):
    global src
    assert src.count(a) == 1, a[:60]
    src = src.replace(a, b)


rep("                    chol_trinv_wave(S, X, stage, lane, &ok);",
    "#ifndef GP_X_NODIAG\n                    chol_trinv_wave(S, X, stage, lane, &ok);\n#endif")
rep("            panel_gemm_t(acc, K + (size_t)(j * T) * ld, ld, own, 0, j, p.nb_max < 0, stage, tid, lane);",
    "#ifndef GP_X_NOLOOP\n            panel_gemm_t(acc, K + (size_t)(j * T) * ld, ld, own, 0, j, p.nb_max < 0, stage, tid, lane);\n#endif")
rep("            panel_gemm_t(acc, L + (size_t)(c * T) * ld, ld, own, a0, c, a, stage, tid, lane);",
    "#ifndef GP_X_NOLOOP\n            panel_gemm_t(acc, L + (size_t)(c * T) * ld, ld, own, a0, c, a, stage, tid, lane);\n#endif")
rep("__device__ __forceinline__ void left_mul_lds(const double* Dl, const v4d (&in)[4][4], int r, int g, F f) {\n",
    "__device__ __forceinline__ void left_mul_lds(const double* Dl, const v4d (&in)[4][4], int r, int g, F f) {\n"
    "#ifdef GP_X_NOLEFT\n#pragma unroll\n    for (int m2 = 0; m2 < 4; ++m2) f(m2, in[m2]);\n    return;\n#endif\n")
rep("                        acc[m][n][e] = K[(size_t)(i * T + 16 * n + r) * ld + j * T + 16 * m + 4 * e + g];",
    "#ifdef GP_X_NOINIT\n                        acc[m][n][e] = 0.0;\n#else\n"
    "                        acc[m][n][e] = K[(size_t)(i * T + 16 * n + r) * ld + j * T + 16 * m + 4 * e + g];\n#endif")
out = os.path.join(ROOT, "scratch", "gpx")
os.makedirs(out, exist_ok=True)
open(os.path.join(out, "ste_gp_x.hip"), "w").write(src)
objs = [os.path.join(ROOT, "ship-track-estimators_amd", "lib", "obj", f) for f in ("ste_kernels.hip.o", "ste_prep.hip.o")]
for v in ("BASE", "NODIAG", "NOLEFT", "NOINIT", "NOLOOP"):
    o = os.path.join(out, "gp_%s.o" % v)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=fast-honor-pragmas",
                    "-DGP_X_" + v, "-c", os.path.join(out, "ste_gp_x.hip"), "-o", o], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", o] + objs + ["-o", os.path.join(out, "libste_%s.so" % v)],
                   check=True)
    print("built", v)
