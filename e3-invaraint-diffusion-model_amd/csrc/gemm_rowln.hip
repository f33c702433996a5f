// BertSelfOutput / BertOutput as ONE launch at large M (transformers 4.38.2 modeling_bert.py; the reference runs them through
// structure_model/model.py:197-213, sequence_model/model.py:226-231):
//     out[M,768] = LayerNorm(A[M,K] . W[768,K]^T + bias + residual) * gamma + beta
// A workgroup owns WHOLE 768-wide rows, so bias + residual + LayerNorm finish in its accumulator registers: the pre-norm sum
// is never written to HBM and never read back (the unfused pair moves 4 x M x 3 KB per site, this kernel 3 x: A in, residual
// in, out).  Arithmetic: the 2-term split products of gemm_split.hip (bf16x3 / f16x3) in the same order (the pre-norm sums
// are bit-identical to that kernel's), fp32 accumulation, the two-pass centred LayerNorm statistics of rowops.hip.
//
// Tile <= 96 rows x 768 columns x 16, 8 waves side by side along N, each wave 96 x 96 = 3 x 3 MFMA tiles of 32x32 (144
// accumulator registers; a 128-row tile needs 192 and hipcc spills ~800 B/lane placing twelve 16-register tuples in 256).
// A workgroup walks ONE contiguous group of rows (256 at M = 65536: tiles of 96 + 96 + 64 rows, m-blocks beyond a tile's
// rows are skipped), so every CU gets the same work whatever M / 96 is.
// Both operands reach LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write):
//   * W is PRE-SPLIT once per weight version into 16-bit hi / lo planes in MFMA-fragment order (e3d_weight_planes_f32_split:
//     [k16 step][32-column block][plane][lane] x 16 B), so one k16 step of the whole weight is 48 contiguous KB that land in
//     LDS as they lie, and a B fragment is one conflict-free lane-linear ds_read_b128;
//   * A stays fp32 in memory; a k32 pair of the tile's rows (128 B per row: whole cache lines) is copied into an image of
//     128-byte rows whose 16-byte chunks are XOR-ed by (row >> 1) & 7 on the SOURCE side (conflict-free ds_read_b128 of
//     fragment rows), and every wave splits the A fragments in registers right before its MFMAs.
// Two W buffers (k16 steps, wave-private: refilled two steps ahead without a barrier) + three A buffers (k32 pairs) = 144 KB;
// the k loop is software-pipelined (the next step's fragments are read under this step's MFMAs into a second register
// set); counted vmcnt waits and ONE raw s_barrier per k32 pair.
// Epilogue: the residual tile streams through a three-buffer LDS ring of 16-row chunks (coalesced 1-KB DMA pieces, never a
// per-lane epilogue load); row sums meet through a transposing shuffle butterfly per half wave and a 3-KB LDS exchange.
#include <type_traits>

#include "e3d_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <typename E> struct V8;
template <> struct V8<__bf16> { typedef bf16x8 t; };
template <> struct V8<_Float16> { typedef f16x8 t; };
__device__ __forceinline__ f32x16 mma16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma16(const f16x8 a, const f16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

constexpr int RN = 768, RBM = 96, NBLK = RN / 32;
constexpr int W_STEP = NBLK * 2 * 1024;          // bytes of one k16 step of the weight planes (49152)
constexpr int A_PAIR = 128 * 128;                // one k32 pair of the A tile: 16 DMA pieces of 8 fp32 rows x 128 B (12 used)
constexpr int LDS_A = 2 * W_STEP;                // three A buffers behind the two W buffers (96 + 48 = 144 KB)
constexpr int RES_CHUNK = 16 * RN * 4;           // residual ring: 16 rows per chunk (49152)
constexpr int LDS_STATS = 3 * RES_CHUNK;         // [8][96] partial sums + [96] row constants, behind everything else
constexpr int LDS_COLS = LDS_STATS + (8 * RBM + RBM) * 4;   // bias | gamma | beta, 3 x 768 floats
constexpr int LDS_TOTAL = LDS_COLS + 3 * RN * 4;            // 160128 of the CU's 163840 (k loop: LDS_A + 3 A_PAIR = 147456)

// the two 16-bit terms of 8 consecutive fp32 values, exactly as gemm_split.hip's split4<2, E> forms them
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, f16x8& hi, f16x8& lo) {
    e3d_f16x2 h[4], l[4];
    e3d_split2_f16(a[0], a[1], h[0], l[0]);
    e3d_split2_f16(a[2], a[3], h[1], l[1]);
    e3d_split2_f16(b[0], b[1], h[2], l[2]);
    e3d_split2_f16(b[2], b[3], h[3], l[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[2 * j] = h[j][0]; hi[2 * j + 1] = h[j][1];
        lo[2 * j] = l[j][0]; lo[2 * j + 1] = l[j][1];
    }
}
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? a[j] : b[j - 4];
        const __bf16 p = (__bf16)x;
        hi[j] = p;
        lo[j] = (__bf16)(x - (float)p);
    }
}

// ---------------------------------------------------------------- weight planes (once per weight version)
// out[((ks * NB + nb) * 2 + plane) * 1024 + lane * 16 + 2 j] = term_plane(W[nb * 32 + (lane & 31)][ks * 16 + 8 (lane >> 5) + j])
template <typename E>
__global__ __launch_bounds__(256) void weight_planes_kernel(const float* __restrict__ W, int N, int K, unsigned char* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int piece = blockIdx.x * 4 + (threadIdx.x >> 6);      // (ks, nb)
    const int nblk = N / 32;
    if (piece >= (K / 16) * nblk) return;
    const int ks = piece / nblk, nb = piece % nblk;
    const float* src = W + (int64_t)(nb * 32 + (lane & 31)) * K + ks * 16 + 8 * (lane >> 5);
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    typename V8<E>::t hi, lo;
    split8(a, b, hi, lo);
    unsigned char* dst = out + (int64_t)piece * 2048 + lane * 16;
    *reinterpret_cast<typename V8<E>::t*>(dst) = hi;
    *reinterpret_cast<typename V8<E>::t*>(dst + 1024) = lo;
}

// timing-only lab builds (tools/lab/rowln_variants.sh; results are garbage): bit 1 no residual, 2 A always pair 0,
// 4 no W DMA in the k loop, 8 no MFMAs, 16 no LayerNorm / stores
#ifndef ROWLN_LAB
#define ROWLN_LAB 0
#endif
#ifdef ROWLN_STAMPS   // lab builds only (tools/lab/rowln_stamps.py): s_memtime stamps of waves 0 and 4 of one workgroup
__device__ long long rowln_stamps[2][64][8];
__device__ long long rowln_tile_stamps[2][8][8];
__device__ long long rowln_wg_times[512][4];     // per workgroup: s_memrealtime (100 MHz) and s_memtime at entry / exit
#define RSTAMP(step, slot)                                                                                        \
    do {                                                                                                          \
        if (stamp_on && (step) < 64 && lane == 0) rowln_stamps[wid >> 2][(step)][(slot)] = __builtin_readcyclecounter(); \
    } while (0)
#define TSTAMP(slot)                                                                                              \
    do {                                                                                                          \
        if (stamp_on && tile_no < 8 && lane == 0) rowln_tile_stamps[wid >> 2][tile_no][(slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define RSTAMP(step, slot) do {} while (0)
#define TSTAMP(slot) do {} while (0)
#endif
#define E3D_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// raw barrier (a __syncthreads() would drain every LDS-DMA in flight), fenced for the COMPILER on both sides
#define E3D_BARRIER()                          \
    do {                                       \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_s_barrier();          \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
// a barrier that publishes this wave's ds_writes (and waits for its ds_reads) without touching the DMA queue
#define E3D_LDS_BARRIER()                                   \
    do {                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
        E3D_BARRIER();                                      \
    } while (0)

// one LDS-DMA instruction: 64 lanes x 16 bytes from g + OFF (per lane) to lds_wave_base + OFF + 16 lane (the immediate
// offset of the instruction applies to both addresses)
template <int OFF = 0>
__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, OFF, 0);
}

// rows of the next tile of a workgroup that has ``left`` rows to go (multiples of 32): 96 while at least 160 remain (or
// exactly 96), otherwise the rest in halves no larger than 96 (256 -> 96, 96, 64; 128 -> 64, 64; 224 -> 96, 64, 64).
// ``pattern`` (workgroup-dependent, 0..3) permutes the order for a 256-row group -- 96 96 64 / 64 96 96 / 96 64 96 /
// 64 64 64 64 -- so that neighbouring workgroups reach their epilogues (residual read + output store: HBM-bound, the matrix
// pipe idle) at different times instead of all CUs hammering HBM together and all running MFMAs together.
__device__ __forceinline__ int next_tile_rows(int left, int pattern, int group_rows) {
#ifndef ROWLN_NO_PATTERNS
    if (group_rows == 256) {
        if (pattern == 1 && left == 256) return 64;
        if (pattern == 2 && left == 160) return 64;
        if (pattern == 3) return 64;
    }
#endif
    if (left >= 160 || left == 96) return 96;
    if (left > 96) return 64;
    return left;
}

template <typename E>
__global__ __launch_bounds__(512, 2) void gemm_rowln_kernel(const float* __restrict__ A, int64_t lda,
                                                            const unsigned char* __restrict__ Wp, const float* __restrict__ bias,
                                                            const float* __restrict__ res, int64_t ldr,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, float* __restrict__ out, int64_t ldo, int M, int K,
                                                            int rows_per_wg, float out_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename V8<E>::t X8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // = this wave's 96-column slice
    const int l31 = lane & 31, half = lane >> 5;
    const int npairs = K / 32, last_step = 2 * npairs - 1;

    int row0 = blockIdx.x * rows_per_wg;
    const int row_end = min(M, row0 + rows_per_wg);
    if (row0 >= row_end) return;

    // ---- DMA sources: a wave-uniform base plus a 32-bit per-lane byte offset; LDS destinations are wave-uniform
    // W: pieces wid * 6 + i of the step's 48 one-KB pieces -- source and destination share the offset inside the step
    const unsigned w_lane = (unsigned)(wid * 6144 + lane * 16);
    // A: pieces wid and wid + 8 (8 rows each) of the pair's 16: lane -> row piece * 8 + (lane >> 3) (clamped to the tile:
    // every wave always issues both pieces, so the counted waits below are the same for all), LDS slot lane & 7 holds the
    // row's 16-byte chunk (lane & 7) ^ ((row >> 1) & 7)
    unsigned a_lane[2];
    auto a_offsets = [&](int rows) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wid + 8 * i) * 8 + (lane >> 3);
            a_lane[i] = (unsigned)(min(row, rows - 1) * (int)lda + (((lane & 7) ^ ((row >> 1) & 7)) << 2)) * 4u;
        }
    };
    // residual: pieces wid * 6 + i of a 16-row chunk's 48 = thirds 0..2 of its rows 2 wid and 2 wid + 1
    unsigned r_lane[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) r_lane[j] = (unsigned)((2 * wid + j) * (int)ldr + lane * 4) * 4u;
    // A fragment reads: row l31 of m-block m, chunk 4 ks + 2 half + j -- the swizzle acts on address bits 4..6 only, so the
    // chunk's address is the row's (2 half) ^ rx slot XOR-ed with the constants 64 ks and 16 j
    const int a_roff = LDS_A + l31 * 128 + (((2 * half) ^ ((l31 >> 1) & 7)) << 4);      // m-block m: + m * 4096
    const int b_roff = (wid * 3) * 2048 + lane * 16;

    // k16 step ``src`` of the weight into W buffer ``buf``
    auto issue_w = [&](int src, int buf) {
        const unsigned char* g = Wp + (int64_t)src * W_STEP + w_lane;
        unsigned char* l = smem + buf * W_STEP + wid * 6144;
        glds16<0>(g, l);
        glds16<1024>(g, l);
        glds16<2048>(g, l);
        glds16<3072>(g, l);
        glds16<0>(g + 4096, l + 4096);
        glds16<1024>(g + 4096, l + 4096);
    };
    // k32 pair ``pair`` of the tile's rows into A buffer ``buf``
    auto issue_a = [&](const float* a_tile, int pair, int buf) {
        const unsigned char* g = reinterpret_cast<const unsigned char*>(a_tile + pair * 32);
        unsigned char* l = smem + LDS_A + buf * A_PAIR + wid * 1024;
        glds16<0>(g + a_lane[0], l);
        glds16<0>(g + a_lane[1], l + 8192);
    };

    float* red = reinterpret_cast<float*>(smem + LDS_STATS);            // [8][96]
    float* rowc = red + 8 * RBM;                                         // [96]
    const int colbase = wid * 96 + l31;
    // per-column constants: parked in LDS once (an ordinary global load next to LDS-DMAs in flight makes hipcc wait for ALL
    // of them, so none is issued inside the tile loop; nine more live registers per lane cost spills)
    float* colc = reinterpret_cast<float*>(smem + LDS_COLS);
    for (int i = tid; i < RN; i += 512) {
        colc[i] = bias ? bias[i] : 0.f;
        colc[RN + i] = gamma[i];
        colc[2 * RN + i] = beta[i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (published by the first tile's opening barrier)
    constexpr float inv_n = 1.0f / RN;
    const bool bit8 = (lane & 8) != 0, bit4 = (lane & 4) != 0, bit2 = (lane & 2) != 0,
               bit1 = (lane & 1) != 0;

    const int pattern = (blockIdx.x >> 3) & 3, group_rows = row_end - row0;
    int rows = next_tile_rows(row_end - row0, pattern, group_rows);
    a_offsets(rows);
    issue_a(A + (int64_t)row0 * lda, 0, 0);
    issue_w(0, 0);
    issue_w(1, 1);
    issue_a(A + (int64_t)row0 * lda, 1, 1);
    int next_row0 = 0, next_rows = 0;
#ifdef ROWLN_STAMPS
    const bool stamp_on = blockIdx.x == 77 && (wid & 3) == 0;
    int tile_no = 0;
    if (tid == 0 && blockIdx.x < 512) {
        rowln_wg_times[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
        rowln_wg_times[blockIdx.x][1] = __builtin_readcyclecounter();
    }
#endif

    // One tile of NM m-blocks (32 NM rows): the whole body is instantiated per NM so that no accumulator ever meets a
    // control-flow merge (hipcc answers those with copies of 16-register tuples, i.e. with spills at this register count).
    // Returns whether the workgroup has another tile; its first loads are then in flight.
    auto tile_body = [&](auto nm_c) -> bool {
        constexpr int NM = decltype(nm_c)::value;
        const float* a_tile = A + (int64_t)row0 * lda;
        f32x16 acc[NM][3];
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

        TSTAMP(0);
        // Software-pipelined k loop.  At the top of step s the wave already holds the step's B fragments (register set s & 1)
        // and the raw A fragment of its first row block: they were read while step s - 1 computed.  So the step opens with
        // MFMAs, W buffer s & 1 is free at once (the wave's own reads of it are a step old: W(s+2) goes in straight away,
        // two full steps ahead), and one third into the step -- behind the first row block's nine MFMAs -- the wave waits for
        // W(s+1) (issued a step ago), reads the next step's fragments into the other register set and carries on.
        // A pairs live in THREE buffers: the workgroup's one barrier per pair sits in the middle of the odd step, after it
        // pair p+1 is visible to everybody (first read: the fragment prefetch that follows) and pair p+2 may be fetched into
        // the buffer pair p-1 was read from (every wave has finished step 2p by then).
        int a_cur = 0, a_nxt = A_PAIR, a_nx2 = 2 * A_PAIR;      // byte offsets of the three A buffers from LDS_A (a_roff carries LDS_A)
        E3D_VMCNT(8);      // A(0) and W(0) have landed (W(1) and A(1), issued after them, may still fly)
        E3D_BARRIER();
        TSTAMP(1);
        X8 bh0[3], bl0[3], bh1[3], bl1[3];
        f32x4 x0, x1;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            bh0[n] = *reinterpret_cast<const X8*>(smem + b_roff + n * 2048);
            bl0[n] = *reinterpret_cast<const X8*>(smem + b_roff + n * 2048 + 1024);
        }
        x0 = *reinterpret_cast<const f32x4*>(smem + a_cur + a_roff);
        x1 = *reinterpret_cast<const f32x4*>(smem + a_cur + (a_roff ^ 16));

        const bool res_on = res != nullptr && !(ROWLN_LAB & 1);
        const unsigned char* res_tile = reinterpret_cast<const unsigned char*>(res + (int64_t)row0 * ldr);
        const int64_t res_chunk_stride = 16 * ldr * 4;
        // step of parity KS (compile time): fragments in (bh, bl); the next step's go to (nh, nl)
        auto step = [&](auto ks_c, X8 (&bh)[3], X8 (&bl)[3], X8 (&nh)[3], X8 (&nl)[3], int w_next, int a_pair_next, int tail_chunk, int stamp_step) {
            constexpr int KS = decltype(ks_c)::value;
            RSTAMP(stamp_step, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the reads of W buffer KS are a step old)
            // the six pieces this step sends to W buffer KS: step s + 2 of the weight -- or, in a tile's last two steps (no
            // such step: tail_chunk = 0 / 1), the first two 16-row chunks of the RESIDUAL, whose ring buffers 0 / 1 are these
            // very regions (a chunk's pieces 6 w .. 6 w + 5 = thirds of its rows 2 w, 2 w + 1): the epilogue finds them landed
            const bool tail = tail_chunk >= 0;
            const unsigned char* gbase = tail ? res_tile + tail_chunk * res_chunk_stride : Wp + (int64_t)w_next * W_STEP;
            const unsigned char* g0 = gbase + (tail ? r_lane[0] : w_lane);
            const unsigned char* g1 = gbase + (tail ? r_lane[1] : w_lane + 3072u);
            unsigned char* lw = smem + KS * W_STEP + wid * 6144;
            f32x4 y0 = x0, y1 = x1;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                X8 ah, al;
                split8(x0, x1, ah, al);
                if (m + 1 < NM) {       // the next row block's fragment flies under this block's nine MFMAs
                    x0 = *reinterpret_cast<const f32x4*>(smem + a_cur + ((a_roff ^ (64 * KS)) + (m + 1) * 4096));
                    x1 = *reinterpret_cast<const f32x4*>(smem + a_cur + ((a_roff ^ (64 * KS + 16)) + (m + 1) * 4096));
                }
#pragma unroll
                for (int n = 0; n < 3; ++n) {     // smallest terms first (the order of gemm_split.hip: bit-identical sums)
                    if (ROWLN_LAB & 8) {
                        asm volatile("" :: "v"(ah), "v"(al), "v"(bh[n]), "v"(bl[n]));
                    } else {
                        acc[m][n] = mma16(ah, bl[n], acc[m][n]);
                        acc[m][n] = mma16(al, bh[n], acc[m][n]);
                        acc[m][n] = mma16(ah, bh[n], acc[m][n]);
                    }
                    if (m == 0) {       // the six W pieces of step s + 2, two behind each triplet of the first row block
                        if (n == 0) { glds16<0>(g0, lw); glds16<1024>(g0, lw); }
                        if (n == 1) { glds16<2048>(g0, lw); glds16<0>(g1, lw + 3072); }
                        if (n == 2) { glds16<1024>(g1, lw + 3072); glds16<2048>(g1, lw + 3072); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (m == 0) {
                    RSTAMP(stamp_step, 1);
                    if (KS == 1) {
                        E3D_VMCNT(12);     // A(p+1) has landed (issued a pair ago; W(2p+2), W(2p+3) behind it may fly)
                        E3D_BARRIER();
                        RSTAMP(stamp_step, 4);
                        if (a_pair_next >= 0) {     // (the last pair fetches nothing: its buffer is the residual ring's third)
                            const unsigned char* ga = reinterpret_cast<const unsigned char*>(a_tile + a_pair_next * 32);
                            unsigned char* la = smem + LDS_A + a_nx2 + wid * 1024;
                            glds16<0>(ga + a_lane[0], la);
                            glds16<0>(ga + a_lane[1], la + 8192);
                        }
                    }
                    E3D_VMCNT(8);          // W(s+1) has landed (issued a step ago); the 6 + 2 pieces behind it may fly
                    RSTAMP(stamp_step, 3);
                    const unsigned char* wn = smem + (KS ^ 1) * W_STEP + b_roff;
#pragma unroll
                    for (int n = 0; n < 3; ++n) {
                        nh[n] = *reinterpret_cast<const X8*>(wn + n * 2048);
                        nl[n] = *reinterpret_cast<const X8*>(wn + n * 2048 + 1024);
                    }
                    const int an = KS == 0 ? a_cur : a_nxt;       // step s + 1: same pair (k16 half 1) / next pair (half 0)
                    y0 = *reinterpret_cast<const f32x4*>(smem + an + (a_roff ^ (64 * (KS ^ 1))));
                    y1 = *reinterpret_cast<const f32x4*>(smem + an + (a_roff ^ (64 * (KS ^ 1) + 16)));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            x0 = y0;
            x1 = y1;
            RSTAMP(stamp_step, 2);
        };

        for (int p = 0; p < npairs; ++p) {
            const bool last_pair = p + 1 == npairs;
            step(std::integral_constant<int, 0>{}, bh0, bl0, bh1, bl1, min(2 * p + 2, last_step), 0, last_pair && res_on ? 0 : -1, 2 * p);
            step(std::integral_constant<int, 1>{}, bh1, bl1, bh0, bl0, min(2 * p + 3, last_step), last_pair ? -1 : min(p + 2, npairs - 1),
                 last_pair && res_on ? 1 : -1, 2 * p + 1);
            const int t = a_cur;
            a_cur = a_nxt;
            a_nxt = a_nx2;
            a_nx2 = t;
        }
        if (!res_on) E3D_VMCNT(0);     // the re-fetched W steps of the tail (with a residual: its chunks 0 and 1 -- let them fly)
        E3D_LDS_BARRIER();
        TSTAMP(2);

        // ------------------------------------------------------------------ epilogue
        // z = fma(acc, out_scale, bias) + residual, exactly as the unfused pair forms it
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = fmaf(acc[m][n][r], out_scale, colc[colbase + 32 * n]);
        if (res_on) {
            // ring of three 16-row chunks: chunk c = tile rows [16 c, 16 c + 16) = m-block c >> 1, accumulator registers
            // 8 (c & 1) .. +8 of every wave's three column blocks; chunks 0 and 1 were requested by the k loop's last two steps
            auto issue_r = [&](int c) {
                const unsigned char* g = res_tile + c * res_chunk_stride;
                unsigned char* l = smem + (c % 3) * RES_CHUNK + wid * 6144;
                glds16<0>(g + r_lane[0], l);
                glds16<1024>(g + r_lane[0], l);
                glds16<2048>(g + r_lane[0], l);
                glds16<0>(g + r_lane[1], l + 3072);
                glds16<1024>(g + r_lane[1], l + 3072);
                glds16<2048>(g + r_lane[1], l + 3072);
            };
            constexpr int NCH = 2 * NM;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c + 1 < NCH) E3D_VMCNT(6);
                else E3D_VMCNT(0);
                E3D_BARRIER();
                if (c + 2 < NCH) issue_r(c + 2);
                const int m = c >> 1, r0 = 8 * (c & 1);
                const unsigned char* cb = smem + (c % 3) * RES_CHUNK + colbase * 4 + half * (4 * RN * 4);
#pragma unroll
                for (int n = 0; n < 3; ++n)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int lr = (j & 3) + 8 * (j >> 2);     // chunk-local row of register r0 + j (half 0)
                        acc[m][n][r0 + j] += *reinterpret_cast<const float*>(cb + lr * (RN * 4) + n * 128);
                    }
            }
        }
        E3D_LDS_BARRIER();     // every wave is past its last read of the W / A buffers and of the ring
        TSTAMP(3);

        // the next tile's first loads fly under the LayerNorm arithmetic and the stores
        next_row0 = row0 + 32 * NM;
        const bool more = next_row0 < row_end;
        if (more) {
            next_rows = next_tile_rows(row_end - next_row0, pattern, group_rows);
            a_offsets(next_rows);
            issue_a(A + (int64_t)next_row0 * lda, 0, 0);
            issue_w(0, 0);
            issue_w(1, 1);
            issue_a(A + (int64_t)next_row0 * lda, 1, 1);
        }

        if (ROWLN_LAB & 16) {
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n) asm volatile("" :: "v"(acc[m][n]));
            if (more) E3D_LDS_BARRIER();
            return more;
        }
        // ---- LayerNorm over the 768 columns of each row: two passes (mean, centred variance), as rowops.hip.
        // Transposing butterfly over the 32 lanes of a half wave: 32 per-lane partial row sums (two m-blocks: index i =
        // 16 m + register) take 16 + 8 + 4 + 2 + 1 exchanges and leave lane l with the total of index l; the 16 sums of a
        // single m-block are first added across lanes l, l ^ 16 and then halved the same way: lanes l and l ^ 16 both end
        // with the total of index l & 15.
        // Level 16 (lanes l <-> l ^ 16, rows of 16 lanes): v_permlane16_swap exchanges row 1 of its first operand with row 0
        // of its second (rows 3 / 2 in the upper half wave), so swap(lo, hi) then lo + hi leaves the row-0 lanes with the
        // lo index summed over both rows and the row-1 lanes with the hi index.  Levels 8 .. 1 stay inside a row: DPP
        // row_mirror (l <-> 15 - l: partners differ in bit 3), row_half_mirror (7 - l: bit 2), quad_perm xor 2 / xor 1 --
        // no LDS round trips (the ds_bpermute form of this butterfly cost ~5 k cycles per pass in waits).
        auto swap16_sum = [&](float lo, float hi) -> float {
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
            return __uint_as_float(r[0]) + __uint_as_float(r[1]);
        };
        auto dpp_xchg = [&](float lo, float hi, bool up, auto ctrl_c) -> float {     // (up ? hi : lo) + the partner's
            constexpr int CTRL = decltype(ctrl_c)::value;
            const float send = up ? lo : hi, keep = up ? hi : lo;
            return keep + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(send), CTRL, 0xF, 0xF, true));
        };
        constexpr int ROW_MIRROR = 0x140, ROW_HALF_MIRROR = 0x141, QUAD_XOR2 = 0x4E, QUAD_XOR1 = 0xB1;
        auto reduce_in_row = [&](float (&w)[16]) -> float {      // 16 values per lane -> lane l: total of index l & 15 over its row
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = dpp_xchg(w[j], w[j + 8], bit8, std::integral_constant<int, ROW_MIRROR>{});
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = dpp_xchg(w[j], w[j + 4], bit4, std::integral_constant<int, ROW_HALF_MIRROR>{});
#pragma unroll
            for (int j = 0; j < 2; ++j) w[j] = dpp_xchg(w[j], w[j + 2], bit2, std::integral_constant<int, QUAD_XOR2>{});
            return dpp_xchg(w[0], w[1], bit1, std::integral_constant<int, QUAD_XOR1>{});
        };
        auto reduce32 = [&](float (&v)[32]) -> float {           // lane l ends with the total of index l (over its half wave)
            float w[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) w[j] = swap16_sum(v[j], v[j + 16]);
            return reduce_in_row(w);
        };
        auto reduce16 = [&](float (&u)[16]) -> float {           // lanes l, l ^ 16 both end with the total of index l & 15
            float w[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) w[j] = swap16_sum(u[j], u[j]);
            return reduce_in_row(w);
        };
        constexpr bool PAIR = NM >= 2, SINGLE = NM != 2;       // a pair of m-blocks (0, 1) and / or a single one (NM - 1)
        constexpr int MS = NM - 1;
        // rows this lane holds after the butterflies (tile-local)
        const int row_v = 32 * (l31 >> 4) + mfma32_row(l31 & 15, half), row_u = 32 * MS + mfma32_row(l31 & 15, half);
        auto exchange = [&](float tv, float tu, bool variance) {
            if (PAIR) red[wid * RBM + row_v] = tv;
            if (SINGLE && l31 < 16) red[wid * RBM + row_u] = tu;
            E3D_LDS_BARRIER();
            if (tid < 32 * NM) {
                const float t = ((red[tid] + red[RBM + tid]) + (red[2 * RBM + tid] + red[3 * RBM + tid])) +
                                ((red[4 * RBM + tid] + red[5 * RBM + tid]) + (red[6 * RBM + tid] + red[7 * RBM + tid]));
                rowc[tid] = variance ? 1.0f / sqrtf(t * inv_n + eps) : t * inv_n;
            }
            E3D_LDS_BARRIER();
        };
        {
            float v[32], u[16], tv = 0.f, tu = 0.f;
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t = (acc[m][0][r] + acc[m][1][r]) + acc[m][2][r];
                    if (PAIR && m < 2) v[m * 16 + r] = t;
                    else u[r] = t;
                }
            if (PAIR) tv = reduce32(v);
            if (SINGLE) tu = reduce16(u);
            exchange(tv, tu, false);
        }
        {
            float v[32], u[16], tv = 0.f, tu = 0.f;
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 mean = *reinterpret_cast<const f32x4*>(rowc + m * 32 + 8 * q + 4 * half);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = 0.f;
#pragma unroll
                        for (int n = 0; n < 3; ++n) {
                            const float d = acc[m][n][4 * q + j] - mean[j];
                            acc[m][n][4 * q + j] = d;
                            t = fmaf(d, d, t);
                        }
                        if (PAIR && m < 2) v[m * 16 + 4 * q + j] = t;
                        else u[4 * q + j] = t;
                    }
                }
            if (PAIR) tv = reduce32(v);
            if (SINGLE) tu = reduce16(u);
            exchange(tv, tu, true);     // (its first barrier also orders every lane's mean reads before the table is rewritten)
        }
        TSTAMP(4);
        float gv[3], bt[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            gv[n] = colc[RN + colbase + 32 * n];
            bt[n] = colc[2 * RN + colbase + 32 * n];
        }
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 rstd = *reinterpret_cast<const f32x4*>(rowc + m * 32 + 8 * q + 4 * half);
                float* o = out + (int64_t)(row0 + m * 32 + 8 * q + 4 * half) * ldo + colbase;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int n = 0; n < 3; ++n) o[(int64_t)j * ldo + 32 * n] = fmaf(acc[m][n][4 * q + j] * rstd[j], gv[n], bt[n]);
            }
        TSTAMP(5);
#ifdef ROWLN_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TSTAMP(6);
        ++tile_no;
#endif
        if (more) E3D_LDS_BARRIER();     // the row constants are read before the next tile's epilogue can rewrite them
        return more;
    };

    while (true) {
        bool more;
        if (rows == 96) more = tile_body(std::integral_constant<int, 3>{});
        else if (rows == 64) more = tile_body(std::integral_constant<int, 2>{});
        else more = tile_body(std::integral_constant<int, 1>{});
        if (!more) break;
        row0 = next_row0;
        rows = next_rows;
    }
    E3D_VMCNT(0);
#ifdef ROWLN_STAMPS
    if (tid == 0 && blockIdx.x < 512) {
        rowln_wg_times[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime();
        rowln_wg_times[blockIdx.x][3] = __builtin_readcyclecounter();
    }
#endif
}

template <typename E>
int launch_rowln(const float* A, int64_t lda, const void* Wp, const float* bias, const float* res, int64_t ldr,
                 const float* gamma, const float* beta, float eps, float* out, int64_t ldo, int M, int K, float out_scale,
                 hipStream_t s) {
    static std::atomic<uint64_t> lds_ok{0};
    e3d_allow_lds(lds_ok, gemm_rowln_kernel<E>, (size_t)LDS_TOTAL);
    // one contiguous row group per workgroup, one workgroup per CU: whole 32-row blocks, as even as M allows
    const int n_cu = e3d_cu_count();
    const int rows_per_wg = ((M + n_cu - 1) / n_cu + 31) / 32 * 32;
    const int grid = (M + rows_per_wg - 1) / rows_per_wg;
    hipLaunchKernelGGL((gemm_rowln_kernel<E>), dim3(grid), dim3(512), LDS_TOTAL, s, A, lda,
                       reinterpret_cast<const unsigned char*>(Wp), bias, res, ldr, gamma, beta, eps, out, ldo, M, K, rows_per_wg,
                       out_scale);
    return e3d_launch_status("e3d_gemm_residual_layernorm_f32_split");
}

}  // namespace

#ifdef ROWLN_STAMPS
extern "C" int e3d_debug_rowln_stamps(long long* steps, long long* tiles) {
    int rc = (int)hipMemcpyFromSymbol(steps, HIP_SYMBOL(rowln_stamps), sizeof(long long) * 2 * 64 * 8);
    if (rc) return rc;
    return (int)hipMemcpyFromSymbol(tiles, HIP_SYMBOL(rowln_tile_stamps), sizeof(long long) * 2 * 8 * 8);
}
extern "C" int e3d_debug_rowln_wg_times(long long* t) {
    return (int)hipMemcpyFromSymbol(t, HIP_SYMBOL(rowln_wg_times), sizeof(long long) * 512 * 4);
}
#endif

extern "C" int64_t e3d_weight_planes_bytes(int N, int K) { return (N % 32 || K % 16 || N <= 0 || K <= 0) ? -1 : (int64_t)N * K * 4; }

extern "C" int e3d_weight_planes_f32_split(const float* W, int N, int K, int terms, void* planes, void* stream) {
    E3D_REQUIRE(W && planes, "weight_planes: null pointer");
    E3D_REQUIRE(N > 0 && K > 0 && N % 32 == 0 && K % 16 == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)planes % 16) == 0,
                "weight_planes: need N%%32==0, K%%16==0, 16-byte alignment (N=%d K=%d)", N, K);
    E3D_REQUIRE(terms == 3 || terms == E3D_TERMS_F16X3, "weight_planes: terms must be 3 or 19 (got %d)", terms);
    const int pieces = (K / 16) * (N / 32);
    const dim3 grid((pieces + 3) / 4), block(256);
    if (terms == 3) hipLaunchKernelGGL(weight_planes_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, W, N, K, (unsigned char*)planes);
    else hipLaunchKernelGGL(weight_planes_kernel<_Float16>, grid, block, 0, (hipStream_t)stream, W, N, K, (unsigned char*)planes);
    return e3d_launch_status("e3d_weight_planes_f32_split");
}

extern "C" int e3d_gemm_residual_layernorm_supported(int M, int N, int K, int64_t lda) {
    return M >= 32 && M % 32 == 0 && N == RN && K >= 64 && K % 32 == 0 && lda >= K && lda % 4 == 0 && (int64_t)RBM * lda < (1ll << 31);
}

extern "C" int e3d_gemm_residual_layernorm_f32_split(const float* A, int64_t lda, const void* w_planes, const float* bias,
                                                     const float* residual, int64_t ldr, const float* gamma, const float* beta,
                                                     float eps, float* out, int64_t ldo, int M, int N, int K, int terms,
                                                     float out_scale, void* stream) {
    E3D_REQUIRE(A && w_planes && gamma && beta && out, "gemm_residual_layernorm: null pointer");
    E3D_REQUIRE(e3d_gemm_residual_layernorm_supported(M, N, K, lda),
                "gemm_residual_layernorm: need M%%32==0, N==768, K%%32==0, K>=64 (M=%d N=%d K=%d lda=%lld)", M, N, K, (long long)lda);
    E3D_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)w_planes % 16) == 0 && ldo >= N, "gemm_residual_layernorm: alignment / ldo");
    E3D_REQUIRE(!residual || (ldr >= N && ldr % 4 == 0 && ((uintptr_t)residual % 16) == 0 && 16 * ldr < (1ll << 31)),
                "gemm_residual_layernorm: residual needs ldr >= N, ldr%%4==0, 16-byte alignment");
    E3D_REQUIRE(terms == 3 || terms == E3D_TERMS_F16X3, "gemm_residual_layernorm: terms must be 3 or 19 (got %d)", terms);
    hipStream_t s = (hipStream_t)stream;
    if (terms == 3) return launch_rowln<__bf16>(A, lda, w_planes, bias, residual, ldr, gamma, beta, eps, out, ldo, M, K, out_scale, s);
    return launch_rowln<_Float16>(A, lda, w_planes, bias, residual, ldr, gamma, beta, eps, out, ldo, M, K, out_scale, s);
}
