// Stand-alone timing of the layer-wise path's product kernels (cnf_lgemm.hip) at BASELINE config 4's shapes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I../../continuousnormalizingflows.jl_amd/csrc lgemm_bench.hip -o lgemm_bench
#include "../../continuousnormalizingflows.jl_amd/csrc/cnf_lgemm.hip"
#include <vector>
using namespace cnf;

static float time_ms(std::function<void()> f, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    const int H = 256; const long long B = 32768;
    float *W, *img, *in, *out, *e, *dout, *slabs, *x, *y;
    hipMalloc(&W, H * (H + 1) * 4); hipMalloc(&img, lg_image_floats(H, H + 1) * 4);
    hipMalloc(&in, (H + 1) * B * 4); hipMalloc(&out, (H + 1) * B * 4); hipMalloc(&e, H * B * 4); hipMalloc(&dout, H * B * 4);
    hipMalloc(&x, H * B * 4); hipMalloc(&y, (H + 1) * B * 4);
    hipMemset(W, 0, H * (H + 1) * 4); hipMemset(in, 0, (H + 1) * B * 4); hipMemset(e, 0, H * B * 4); hipMemset(x, 0, H * B * 4); hipMemset(y, 0, (H + 1) * B * 4);
    long long chunk; const int nch = lg_wgrad_chunks(H, B, 256, &chunk);
    hipMalloc(&slabs, (size_t)nch * H * (H + 1) * 4); hipMemset(slabs, 0, (size_t)nch * H * (H + 1) * 4);
    lg_pack_image(W, 1, H, H, H + 1, img, 0);
    const double gf = 2.0 * H * H * B / 1e9;
    float t;
    t = time_ms([&] { lg_gemm(img, H, H + 1, in, H + 1, out, H + 1, B, LG_EPI_ACT, nullptr, 0, dout, H, CNF_ACT_TANH, 0, nullptr, nullptr, 0, 0); }, 50);
    printf("lg_gemm ACT   256 x 257 x %lld : %.1f us  %.1f TFLOP/s\n", B, t * 1e3, gf / t);
    t = time_ms([&] { lg_gemm(img, H, H, in, H, out, H, B, LG_EPI_PLAIN, nullptr, 0, nullptr, 0, 0, 0, nullptr, nullptr, 0, 0); }, 50);
    printf("lg_gemm PLAIN 256 x 256 x %lld : %.1f us  %.1f TFLOP/s\n", B, t * 1e3, gf / t);
    t = time_ms([&] { lg_gemm(img, H, H, in, H, out, H, B, LG_EPI_MUL2, e, H, dout, H, 0, 0, nullptr, nullptr, 0, 0); }, 50);
    printf("lg_gemm MUL2  256 x 256 x %lld : %.1f us  %.1f TFLOP/s\n", B, t * 1e3, gf / t);
    t = time_ms([&] { lg_wgrad(slabs, (long long)H * (H + 1), chunk, nch, H, H + 1, x, H, y, H + 1, B, 0); }, 50);
    printf("lg_wgrad 256 x 257, %d chunks of %lld : %.1f us  %.1f TFLOP/s\n", nch, chunk, t * 1e3, gf / t);
    t = time_ms([&] { lg_wgrad(slabs, (long long)H * (H + 1), chunk, nch, H, H, x, H, y, H, B, 0); }, 50);
    printf("lg_wgrad 256 x 256 : %.1f us  %.1f TFLOP/s\n", t * 1e3, gf / t);
    t = time_ms([&] { lg_gemm(img, H, 33, in, 33, out, H + 1, B, LG_EPI_ACT, nullptr, 0, dout, H, CNF_ACT_TANH, 0, nullptr, nullptr, 0, 0); }, 50);
    printf("lg_gemm ACT   256 x 33 (layer 1) : %.1f us\n", t * 1e3);
    t = time_ms([&] { lg_gemm(img, 32, H, in, H, out, 33, B, LG_EPI_PLAIN, nullptr, 0, nullptr, 0, 0, 0, nullptr, nullptr, 0, 0); }, 50);
    printf("lg_gemm PLAIN 32 x 256 (D rows) : %.1f us\n", t * 1e3);
    return 0;
}
