// Where does lg_wgrad's time go?  The long-K call of the cooperative gradient (256 x 257 cotangent over 262 144 columns, Y with
// ld = 272) built normally and in timing-only forms (-DLG_EXP_NOFETCH: no global loads in the steady-state loop; _NOPARK: no LDS
// writes; _NOSYNC: no barriers - results are wrong in those builds, only the durations mean something).
//   for v in "" -DLG_EXP_NOFETCH -DLG_EXP_NOPARK "-DLG_EXP_NOFETCH -DLG_EXP_NOPARK" "-DLG_EXP_NOFETCH -DLG_EXP_NOPARK -DLG_EXP_NOSYNC"; do
//     hipcc -O3 -std=c++17 --offload-arch=gfx950 $v -I../../include -I../../continuousnormalizingflows.jl_amd/csrc wgrad_variants.hip -o wv && ./wv; done
#include "../../continuousnormalizingflows.jl_amd/csrc/cnf_lgemm.hip"
#include <functional>
using namespace cnf;

static float time_ms(std::function<void()> f, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main(int argc, char** argv) {
    const int H = 256, ldy = 272; const long long B2 = 262144;
    float *slabs, *x, *y;
    hipMalloc(&x, (size_t)H * B2 * 4); hipMalloc(&y, (size_t)ldy * B2 * 4);
    hipMemset(x, 0, (size_t)H * B2 * 4); hipMemset(y, 0, (size_t)ldy * B2 * 4);
    for (int per_cu : {3, 4, 6}) {
        long long chunk; const int nch = lg_wgrad_chunks(H, B2, 256, &chunk, per_cu);
        hipMalloc(&slabs, (size_t)nch * H * (H + 1) * 4); hipMemset(slabs, 0, (size_t)nch * H * (H + 1) * 4);
        const double gf = 2.0 * H * (H + 1) * B2 / 1e9;
        float t = time_ms([&] { lg_wgrad(slabs, (long long)H * (H + 1), chunk, nch, H, H + 1, x, H, y, ldy, B2, 0); }, 200);
        printf("%s per_cu %d: %d chunks of %lld : %.1f us  %.1f TFLOP/s\n", argc > 1 ? argv[1] : "", per_cu, nch, chunk, t * 1e3, gf / t);
        hipFree(slabs);
    }
    return 0;
}
