// cnf_wgrad_tiles.hip - weight cotangents from TILE-NATIVE operands (round 6):
//     C[M x Nc] += sum over column tiles ct of  X0(ct) Y0(ct)^T + X1(ct) Y1(ct)^T          (samples on the MFMA K axis)
// the two terms of  Wbar_{l+1} += delta_{l+1} vbar_l^T + sbar_{l+1} h_l^T  (reference: the pullback of a Dense layer under
// Zygote, reached through QuadratureAdjoint + ZygoteVJP, src/core/icnf.jl:90-99), with delta / h straight from the forward
// solve's stage store and vbar / sbar from the reverse sweep (cnf_tiles.h).  The bias cotangent (row sums of sbar) rides along.
//
// What differs from lg_wgrad (cnf_lgemm.hip), whose operands are column-major arrays written through an LDS transposition by
// the sweep: a tile arrives with ONE 16-byte load per lane (the producer's own layout), is parked in LDS with one
// ds_write_b128 per lane into a copy whose four lane groups sit 72 words apart, and is read back as MFMA operands - lane
// (j = lane & 15, kq = lane >> 4), k-step u: feature j of sample 4 u + kq - with ds_read_b32 at
// (j & 3) * 72 + (4 u + kq) * 4 + (j >> 2): the 32 lanes of a half-wave hit 32 different banks ((j & 3) * 8 + kq * 4 + (j >> 2),
// kq in {0, 1}).  The row operand needs no LDS at all: every wave owns one 16-row strip of C and reads its X tile from global
// memory directly in operand form (four dword loads per tile, 64-byte segments).  Tile offsets and k-steps are immediates of
// the LDS instructions: the steady-state loop carries no address arithmetic.
#include "cnf_tiles.h"
#include "cnf_mfma_dev.h"

namespace cnf {

namespace {

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int kTileWords = 4 * 72;   // a parked tile: four lane groups, 64 + 8 words each
// 16-column tiles of C a wave keeps at most: a strip of up to 12 tiles (H <= 192: the default architecture up to nvariables = 23) is ONE
// column group - its X operand is read once, and an iteration carries 4 NTN MFMAs per wave against a fixed cost (a barrier, the
// fetches, the parks); wider strips are dealt evenly over groups of <= 12 (16 tiles = 8 + 8)
constexpr int kMaxNTN = 12;
// (Measured and dropped, round 6 - PMC of this kernel at cfg4: 67 % MFMA-busy, waves waiting 31 % of their cycles, no LDS bank conflict,
// HBM traffic = every operand once: four slices in flight instead of two: 0.2195 against 0.2185 ms per 256 x 256 call, and at twelve
// tiles per group, where the registers then allow two waves per SIMD instead of three, 0.193 against 0.155 ms - occupancy, not prefetch
// depth, is what hides the round trips; LDS-only barriers instead of __syncthreads: no difference.)

// NTN: 16-column tiles of C a wave keeps (one column group).  Four waves = four 16-row strips of C, sharing every Y tile.
template <int NTN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
wgrad_tiles_kernel(WTArgs a) {
    constexpr int NSL = (NTN + 3) / 4;              // Y tiles a wave fetches per (column tile, term)
    constexpr int BUFW = NSL * 4 * kTileWords;      // words per LDS buffer (room for 4 NSL tiles: parking is unconditional)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroup -> (chunk of column tiles, row block, column group); the sharers of a chunk get ids equal mod 8: one XCD, one L2
    const int sharers = a.rblocks * a.groups;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int chunk = (slot / sharers) * 8 + xcd, sub = slot % sharers;
    if (chunk >= a.nchunks) return;
    const int rblock = sub % a.rblocks, cgp = sub / a.rblocks;
    const int mt = rblock * 4 + wave;                // this wave's 16-row strip of C
    const int ct0 = cgp * NTN;                       // first 16-column tile of this group
    const long long c0 = (long long)chunk * a.chunk;
    const long long c1 = c0 + a.chunk < a.nct ? c0 + a.chunk : a.nct;
    const int nrem = (int)(c1 - c0);
    const bool strip_live = 16 * mt < a.M;
    const int ntl = ((a.Nc + 15) / 16 - ct0) < NTN ? ((a.Nc + 15) / 16 - ct0) : NTN;   // live column tiles of this group (>= 1)

    // buffer resources end behind the chunk's last column tile: a prefetch past the chunk reads zeros, no branch
    __amdgpu_buffer_rsrc_t rX[2], rY[2];
    unsigned xo[2], yo[2][NSL], xst[2], yst[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const WTTerm& t = a.t[k];
        rX[k] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t.x + c0 * t.xtm * 256), 0, (int)((long long)nrem * t.xtm * 1024), 0x00020000);
        rY[k] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t.y + c0 * t.ytm * 256), 0, (int)((long long)nrem * t.ytm * 1024), 0x00020000);
        const int mx = mt < t.xtm ? mt : t.xtm - 1;
        xo[k] = 4u * (unsigned)(mx * 256 + (n & 3) * 64 + g * 4 + (n >> 2));             // + 64 u bytes: k-step u
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            const int ty = ct0 + wave + 4 * s;
            yo[k][s] = 4u * (unsigned)((ty < t.ytm ? ty : t.ytm - 1) * 256 + lane * 4);
        }
        xst[k] = (unsigned)t.xtm * 1024u; yst[k] = (unsigned)t.ytm * 1024u;
    }
    struct Slice { f32x4 yv[NSL]; f32x4 xv; };
    auto fetch = [&](int k, int c, Slice& sl) {       // column tile c (relative to the chunk) of term k
        const unsigned sx = (unsigned)c * xst[k], sy = (unsigned)c * yst[k];
#pragma unroll
        for (int s = 0; s < NSL; ++s) sl.yv[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rY[k], (int)yo[k][s], (int)sy, 0));
#pragma unroll
        for (int u = 0; u < 4; ++u) sl.xv[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX[k], (int)(xo[k] + 64u * u), (int)sx, 0));
    };
    f32x4 xs[2];
    float* pw = smem + (wave * kTileWords + g * 72 + n * 4);          // this lane's slot in tile `wave` of a buffer
    auto park = [&](int buf, const Slice& sl) {
#pragma unroll
        for (int s = 0; s < NSL; ++s) *reinterpret_cast<f32x4*>(pw + buf * BUFW + 4 * s * kTileWords) = sl.yv[s];
        xs[buf] = sl.xv;
    };
    const float* pr = smem + ((n & 3) * 72 + g * 4 + (n >> 2));
    f32x4 acc[NTN];
#pragma unroll
    for (int t = 0; t < NTN; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const bool want_bias = a.bias && cgp == 0;
    auto multiply = [&](int buf) {
        if (!strip_live) return;
        const float* b = pr + buf * BUFW;
        const f32x4 av = xs[buf];
        if (buf == 1 && want_bias) bsum += (av[0] + av[1]) + (av[2] + av[3]);
        float bv[NTN][4];
#pragma unroll
        for (int t = 0; t < NTN; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) bv[t][u] = b[t * kTileWords + 16 * u];
        constexpr int NTM = NTN > 1 ? NTN - 1 : NTN;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < NTM; ++t) acc[t] = mfma4(av[u], bv[t][u], acc[t]);
        if constexpr (NTN > 1) {
            if (ntl >= NTN) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[NTN - 1] = mfma4(av[u], bv[NTN - 1][u], acc[NTN - 1]);
            }
        }
    };
    // C rows 16 mt + 4 g + e, column 16 (ct0 + t) + n: four consecutive floats per lane
    float* C = a.slabs + (long long)chunk * a.slab_stride;
    const int r0 = 16 * mt + 4 * g;
    const bool rows4 = r0 + 3 < a.M;
    f32x4 prev[NTN];
    float bprev = 0.f;
    auto load_prev = [&]() {
#pragma unroll
        for (int t = 0; t < NTN; ++t) {
            const int col = 16 * (ct0 + t) + n;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (col < a.Nc && strip_live) {
                const float* cp = C + (long long)col * a.M + r0;
                if (rows4) { const f32x4u q = *reinterpret_cast<const f32x4u*>(cp); v = f32x4{q[0], q[1], q[2], q[3]}; }
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (r0 + r < a.M) v[r] = cp[r];
                }
            }
            prev[t] = v;
        }
        if (want_bias && g == 0 && 16 * mt + n < a.M) bprev = C[(long long)a.Nc * a.M + 16 * mt + n];
    };

    Slice slA, slB;
    fetch(0, 0, slA);
    park(0, slA);
    fetch(1, 0, slA);
    __builtin_amdgcn_sched_barrier(0);
    fetch(0, 1, slB);
    __syncthreads();
    int c = 0;
    for (; c + 1 < nrem; ++c) {
        multiply(0);                 // (c, term 0); set A holds (c, term 1), set B (c + 1, term 0)
        park(1, slA);
        fetch(1, c + 1, slA);
        __syncthreads();
        multiply(1);                 // (c, term 1)
        park(0, slB);
        fetch(0, c + 2, slB);
        __syncthreads();
    }
    multiply(0);
    park(1, slA);
    __syncthreads();
    load_prev();
    multiply(1);
    if (!strip_live) return;
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
        const int col = 16 * (ct0 + t) + n;
        if (col >= a.Nc) continue;
        float* cp = C + (long long)col * a.M + r0;
        const f32x4 v = prev[t] + acc[t];
        if (rows4) *reinterpret_cast<f32x4u*>(cp) = f32x4u{v[0], v[1], v[2], v[3]};
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (r0 + r < a.M) cp[r] = v[r];
        }
    }
    if (want_bias) {
        // lanes (n, kq = 0 .. 3) hold the partial row sums of row 16 mt + n over the samples 4 u + kq
        float s = bsum;
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (g == 0 && 16 * mt + n < a.M) C[(long long)a.Nc * a.M + 16 * mt + n] = bprev + s;
    }
}

template <int NTN>
hipError_t launch_wt(const WTArgs& a, dim3 grid, hipStream_t st) {
    constexpr int lds = 2 * ((NTN + 3) / 4) * 4 * kTileWords * (int)sizeof(float);
    hipLaunchKernelGGL(wgrad_tiles_kernel<NTN>, grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace

// chunks of column tiles (= slabs) for a C of M x Nc over nct column tiles: three workgroups are resident per CU; six per CU are
// asked for when five or more workgroups share a chunk (uneven strips even out), as lg_wgrad_chunks found for the same loop shape
int wgrad_tiles_chunks(int M, int Nc, long long nct, int num_cus, long long* chunk_out) {
    const int RB = ((M + 15) / 16 + 3) / 4;
    const int groups = ((Nc + 15) / 16 + kMaxNTN - 1) / kMaxNTN;
    const int sharers = RB * groups;
    const int per_cu = sharers >= 5 ? 6 : 3;
    long long want = ((long long)per_cu * num_cus + sharers - 1) / sharers;
    if (want < 1) want = 1;
    long long chunk = (nct + want - 1) / want;
    if (chunk < 4) chunk = 4;
    const long long nch = (nct + chunk - 1) / chunk;
    if (chunk_out) *chunk_out = chunk;
    return (int)(nch < 1 ? 1 : nch);
}

hipError_t wgrad_tiles(float* slabs, long long slab_stride, long long chunk, int nchunks, int M, int Nc, const WTTerm& t0, const WTTerm& t1,
                       long long nct, int bias, hipStream_t st) {
    if (nct <= 0) return hipSuccess;
    WTArgs a{};
    a.slabs = slabs; a.slab_stride = slab_stride; a.t[0] = t0; a.t[1] = t1; a.nct = nct; a.chunk = chunk;
    a.M = M; a.Nc = Nc; a.bias = bias;
    const int MT = (M + 15) / 16, NT = (Nc + 15) / 16;
    const int groups = (NT + kMaxNTN - 1) / kMaxNTN;
    a.nchunks = nchunks; a.rblocks = (MT + 3) / 4; a.groups = groups;
    const dim3 grid((unsigned)(((nchunks + 7) / 8) * 8 * a.rblocks * a.groups));
    const int ntn = (NT + groups - 1) / groups;   // the strip's tiles dealt evenly over its groups
    switch (ntn) {
        case 1: return launch_wt<1>(a, grid, st);
        case 2: return launch_wt<2>(a, grid, st);
        case 3: return launch_wt<3>(a, grid, st);
        case 4: return launch_wt<4>(a, grid, st);
        case 5: return launch_wt<5>(a, grid, st);
        case 6: return launch_wt<6>(a, grid, st);
        case 7: return launch_wt<7>(a, grid, st);
        case 8: return launch_wt<8>(a, grid, st);
        case 9: return launch_wt<9>(a, grid, st);
        case 10: return launch_wt<10>(a, grid, st);
        case 11: return launch_wt<11>(a, grid, st);
        default: return launch_wt<12>(a, grid, st);
    }
}

}  // namespace cnf
