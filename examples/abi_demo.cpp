// abi_demo.cpp — the C ABI of libcnf_hip.so used from a plain C++/HIP host: no Python, no torch.
//
//   hipcc -O2 --offload-arch=gfx950 -I include examples/abi_demo.cpp -o abi_demo \
//         -L continuousnormalizingflows.jl_amd -lcnf_hip -Wl,-rpath,$PWD/continuousnormalizingflows.jl_amd
//   ./abi_demo in.bin out.bin
//
// in.bin:  int32 {nvars, H, L, B, nsteps, alg}, then float32 p[nparams], x[nvars*B], eps[nvars*B]
//          (FFJORD: Dense(nvars+1 => H, tanh), (L-1) x Dense(H => H, tanh), Dense(H => nvars); column-major arrays,
//          one column per sample, exactly what the Julia side would hand over)
// out.bin: float32 logp[B];  stdout: mean log-density and the time of one solve.
// This is the call sequence INTEGRATION.md's Julia glue makes: cnf_create -> cnf_set_params -> cnf_inference_fixed, then the
// default solver: cnf_assemble_u0 -> cnf_solve_vcabm -> cnf_epilogue.  The output file holds both log-density vectors.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cnf.h"

#define HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 2; } } while (0)
#define CNF_OK_(e) do { int _r = (e); if (_r != CNF_OK) { fprintf(stderr, "%s: %d %s\n", #e, _r, cnf_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int hdr[6];
    if (fread(hdr, sizeof(int), 6, f) != 6) return 1;
    const int nvars = hdr[0], H = hdr[1], L = hdr[2], nsteps = hdr[4], alg = hdr[5];
    const long long B = hdr[3];
    cnf_config cfg{};
    cfg.nvars = nvars; cfg.naug = 0; cfg.ncond = 0; cfg.autonomous = 0; cfg.n_layers = L + 1;
    cfg.widths[0] = nvars + 1;
    for (int l = 1; l <= L; ++l) { cfg.widths[l] = H; cfg.acts[l - 1] = CNF_ACT_TANH; }
    cfg.widths[L + 1] = nvars; cfg.acts[L] = CNF_ACT_IDENTITY;
    cfg.mode = CNF_MODE_HUTCH_VJP; cfg.nprobes = 1; cfg.kernel_path = CNF_PATH_AUTO; cfg.arith = CNF_ARITH_F32;
    std::vector<size_t> w_off, b_off;
    size_t np = 0;
    for (int l = 0; l <= L; ++l) {   // Lux layout: weight (out x in, column-major), then bias
        w_off.push_back(np); np += (size_t)cfg.widths[l] * cfg.widths[l + 1];
        b_off.push_back(np); np += (size_t)cfg.widths[l + 1];
    }
    std::vector<float> p(np), x((size_t)nvars * B), eps((size_t)nvars * B), logp(B);
    if (fread(p.data(), 4, np, f) != np || fread(x.data(), 4, x.size(), f) != x.size() ||
        fread(eps.data(), 4, eps.size(), f) != eps.size()) { fprintf(stderr, "short input\n"); return 1; }
    fclose(f);

    float *dx, *de, *dl;
    HIP_OK(hipMalloc((void**)&dx, x.size() * 4)); HIP_OK(hipMalloc((void**)&de, eps.size() * 4)); HIP_OK(hipMalloc((void**)&dl, B * 4));
    HIP_OK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(de, eps.data(), eps.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    cnf_handle* h = nullptr;
    CNF_OK_(cnf_create(&h, &cfg));
    CNF_OK_(cnf_set_params(h, p.data(), np, w_off.data(), b_off.data(), /*p_is_device=*/0, st));
    CNF_OK_(cnf_inference_fixed(h, alg, nsteps, 0.f, 1.f, dx, de, nullptr, B, dl, nullptr, nullptr, st));   // warm-up
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, st));
    CNF_OK_(cnf_inference_fixed(h, alg, nsteps, 0.f, 1.f, dx, de, nullptr, B, dl, nullptr, nullptr, st));
    HIP_OK(hipEventRecord(e1, st));
    HIP_OK(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    HIP_OK(hipMemcpy(logp.data(), dl, B * 4, hipMemcpyDeviceToHost));
    double s = 0;
    for (float v : logp) s += v;
    printf("kernel_path=%d B=%lld mean_logp=%.6f solve_ms=%.3f samples_steps_per_s=%.3e\n", cnf_kernel_path(h), B, s / B, ms,
           (double)B * nsteps / (ms * 1e-3));
    FILE* g = fopen(argv[2], "wb");
    if (!g) { perror(argv[2]); return 1; }
    fwrite(logp.data(), 4, B, g);

    // the reference's default sol_kwargs (alg = VCABM(), reltol = abstol = 1e-4): u0 = [x; 0] -> one cnf_solve_vcabm -> epilogue
    const int S = nvars + 3;
    float *du0, *du1;
    HIP_OK(hipMalloc((void**)&du0, (size_t)S * B * 4)); HIP_OK(hipMalloc((void**)&du1, (size_t)S * B * 4));
    CNF_OK_(cnf_assemble_u0(h, dx, B, du0, st));
    cnf_solve_stats stats{};
    std::vector<float> dts(256);
    std::vector<int32_t> orders(256);
    CNF_OK_(cnf_solve_vcabm(h, 0.f, 1.f, du0, de, nullptr, B, 1e-4f, 1e-4f, 0.f, 100000, du1, &stats, dts.data(), orders.data(), 256, st));
    CNF_OK_(cnf_epilogue(h, du1, B, dl, nullptr, st));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(logp.data(), dl, B * 4, hipMemcpyDeviceToHost));
    printf("vcabm naccept=%d nreject=%d nf=%d max_order=%d first_dt=%.6g\n", stats.naccept, stats.nreject, stats.nf, stats.max_order, dts[0]);
    fwrite(logp.data(), 4, B, g);
    fclose(g);
    hipFree(du0); hipFree(du1);
    CNF_OK_(cnf_destroy(h));
    hipFree(dx); hipFree(de); hipFree(dl);
    return 0;
}
