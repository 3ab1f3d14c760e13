// Diagnostic (not part of libltxk.so): what MFMA rate does the chip sustain from registers alone, per instruction shape and
// operand data?  No LDS, no global traffic inside the timed loop -- the difference between this number and the GEMM's is
// what feeding the matrix cores costs; the difference between random and zero operands is the power limit.
//   hipcc -O3 --offload-arch=gfx950 scripts/mfma_power_probe.hip -o gpurun_out/mfma_power_probe && gpurun_out/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cstring>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// Inline-asm MFMAs accumulate in place in AGPRs (the builtin form lets hipcc shuffle accumulators between registers).
// NA x NB register tile of 16x16x32 MFMAs (NA + NB operand fragments feed NA*NB instructions per K-step)
template <int NA, int NB>
__global__ __launch_bounds__(256) void probe16(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
    bf16x8 a[NA], b[NB];
    const uint4* p = src + (size_t)(blockIdx.x * 256 + threadIdx.x) * (NA + NB);
    for (int i = 0; i < NA; i++) a[i] = __builtin_bit_cast(bf16x8, p[i]);
    for (int i = 0; i < NB; i++) b[i] = __builtin_bit_cast(bf16x8, p[NA + i]);
    f32x4 acc[NA][NB] = {};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NA; i++)
#pragma unroll
            for (int j = 0; j < NB; j++) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
    }
    float s = 0.f;
    for (int i = 0; i < NA; i++) for (int j = 0; j < NB; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NA, int NB>
__global__ __launch_bounds__(256) void probe32(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
    bf16x8 a[NA], b[NB];
    const uint4* p = src + (size_t)(blockIdx.x * 256 + threadIdx.x) * (NA + NB);
    for (int i = 0; i < NA; i++) a[i] = __builtin_bit_cast(bf16x8, p[i]);
    for (int i = 0; i < NB; i++) b[i] = __builtin_bit_cast(bf16x8, p[NA + i]);
    f32x16 acc[NA][NB] = {};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NA; i++)
#pragma unroll
            for (int j = 0; j < NB; j++) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
    }
    float s = 0.f;
    for (int i = 0; i < NA; i++) for (int j = 0; j < NB; j++) for (int k = 0; k < 16; k++) s += acc[i][j][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static uint16_t bf16_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }


// ---- what feeding the matrix cores costs ---------------------------------------------------------------------------------
// One workgroup of 8 waves per CU (2 per SIMD, the GEMM's occupancy), each wave an 80x64 tile of 16x16x32 MFMAs like
// gemm.hip's 160x256 layout: per K-step of 64 that is 40 MFMAs, 2 x R ds_read_b128 fragment reads and G 1-KiB LDS-DMA
// pieces per wave.  R = 9 and G = 6.5 are the GEMM's own ratios; FAR of every 13 DMA pieces come from a buffer larger than
// the L2s (Infinity-Cache / HBM traffic), the rest hit L2.  No barriers, no epilogue: only the steady-state energy mix.
__device__ __forceinline__ void glds16_s(uint64_t base, unsigned lane_off, void* lds_dst) {
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(base), "s"(dst) : "memory");
}

__device__ unsigned long long g_clk[4];        // block 0: s_memtime / s_memrealtime (100 MHz) before and after the loop

// the same piece through the buffer path: base in a 4-SGPR resource (raw buffer, 0x00020000 = gfx9 data format 32 bit), one
// 32-bit VGPR offset; M0 as above
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void blds16(i32x4 rsrc, unsigned off, void* lds_dst) {
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(off), "s"(rsrc), "s"(dst) : "memory", "m0");
}

// the same without saving / restoring M0 around each piece (nothing else in the loop reads M0)
__device__ __forceinline__ void glds16_nosave(uint64_t base, unsigned lane_off, void* lds_dst) {
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane_off), "s"(base), "s"(dst) : "memory", "m0");
}

template <int R, int G2, int FAR, int NA = 5, int NB = 4, int NWAVES = 8, int RU = R, bool SINGLE = false, bool NOSAVE = false, bool BUF = false>     // G2 = DMA pieces per wave per TWO K-steps
__global__ __launch_bounds__(NWAVES * 64) void feed(const uint4* __restrict__ near_buf, size_t near_bytes, const uint4* __restrict__ far_buf,
                                            size_t far_bytes, float* __restrict__ out, int iters) {
    __shared__ uint4 lds[8192];                       // [0, 64 KiB): fragment source, [64, 128 KiB): DMA sink (8 slots per wave)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4096; i += NWAVES * 64) lds[i] = near_buf[(size_t)blockIdx.x * 4096 % (near_bytes / 16 - 4096) + i];
    __syncthreads();
    constexpr int NF = NA + NB;
    constexpr int NA_A = (NWAVES > 4 && NA * NB * 4 > 128) ? 128 / (NB * 4) : NA;
    bf16x8 f0[NF], f1[NF];
    for (int i = 0; i < NF; i++) { f0[i] = __builtin_bit_cast(bf16x8, lds[(i * 64 + lane) & 4095]); if (!SINGLE) f1[i] = __builtin_bit_cast(bf16x8, lds[(i * 64 + 640 + lane) & 4095]); }
    f32x4 acc[NA][NB] = {};
    // this wave's private windows of the two global buffers
    const size_t nwin = near_bytes / (gridDim.x * NWAVES), fwin = far_bytes / (gridDim.x * NWAVES);
    const uint64_t nbase = (uint64_t)near_buf + (size_t)(blockIdx.x * NWAVES + wave) * nwin;
    const uint64_t fbase = (uint64_t)far_buf + (size_t)(blockIdx.x * NWAVES + wave) * fwin;
    unsigned noff = 0, foff = 0, rd = wave * 97;
    uint4* sink = lds + 4096 + wave * (4096 / NWAVES);
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = __builtin_amdgcn_s_memtime(); g_clk[1] = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {              // 4 sub-steps of K=32 = two K-steps of 64
            // SINGLE: one fragment set per wave - read, wait, multiply; the SIMD's other wave covers the read latency
            bf16x8* cur = SINGLE ? f0 : ((ks & 1) ? f1 : f0);
            bf16x8* nxt = SINGLE ? f0 : ((ks & 1) ? f0 : f1);
            rd = (rd + 577) & 4095;
#pragma unroll
            for (int i = 0; i < R; i++) {
                bf16x8 v = __builtin_bit_cast(bf16x8, lds[(rd + i * 64 + lane) & 4095]);
                if (i < RU) nxt[i] = v; else asm volatile("" :: "v"(v));      // read issued, value not fed to an MFMA
            }
#pragma unroll
            for (int i = 0; i < NA; i++)
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    // hipcc splits a 256-register budget 128 AGPR / 128 VGPR: accumulator rows past 128 registers live in VGPRs
                    if (i < NA_A) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(cur[i]), "v"(cur[NA + j]));
                    else          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(cur[i]), "v"(cur[NA + j]));
                }
        }
#pragma unroll
        for (int g = 0; g < G2; g++) {
            if (BUF) {
                i32x4 rs;
                const uint64_t bb = g < FAR ? fbase : nbase;
                rs[0] = (int)(unsigned)bb; rs[1] = (int)(unsigned)(bb >> 32) & 0xffff; rs[2] = -1; rs[3] = 0x00020000;
                if (g < FAR) { blds16(rs, foff + lane * 16, sink + (g % (64 / NWAVES)) * 64); foff += 1024; if (foff >= fwin) foff = 0; }
                else         { blds16(rs, noff + lane * 16, sink + (g % (64 / NWAVES)) * 64); noff += 1024; if (noff >= nwin) noff = 0; }
                continue;
            }
            if (NOSAVE) {
                if (g < FAR) { glds16_nosave(fbase + foff, lane * 16, sink + (g % (64 / NWAVES)) * 64); foff += 1024; if (foff >= fwin) foff = 0; }
                else         { glds16_nosave(nbase + noff, lane * 16, sink + (g % (64 / NWAVES)) * 64); noff += 1024; if (noff >= nwin) noff = 0; }
                continue;
            }
            if (g < FAR) { glds16_s(fbase + foff, lane * 16, sink + (g % (64 / NWAVES)) * 64); foff += 1024; if (foff >= fwin) foff = 0; }
            else         { glds16_s(nbase + noff, lane * 16, sink + (g % (64 / NWAVES)) * 64); noff += 1024; if (noff >= nwin) noff = 0; }
        }
        if (G2 > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(G2 > 0 ? G2 : 0));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[2] = __builtin_amdgcn_s_memtime(); g_clk[3] = __builtin_amdgcn_s_memrealtime(); }
    float s = 0.f;
    for (int i = 0; i < NA; i++) for (int j = 0; j < NB; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * NWAVES * 64 + threadIdx.x] = s;
}


// W operand straight from global memory into registers (no LDS write, no LDS read for it): each wave loads its own four
// 16x32 W fragments per K-sub-step with global_load_dwordx4, three sub-steps ahead; the two waves that share a W column
// block read the same addresses (the second hits the CU's L1).  A still comes by LDS-DMA (2.5 pieces per wave and K-step)
// and ds_read_b128.  FAR_W of every 16 W loads and FAR_A of every 5 A pieces come from the far window.
template <int FAR_A, int FAR_W>
__global__ __launch_bounds__(512) void feed_direct(const uint4* __restrict__ near_buf, size_t near_bytes, const uint4* __restrict__ far_buf,
                                                   size_t far_bytes, float* __restrict__ out, int iters) {
    __shared__ uint4 lds[8192];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = near_buf[(size_t)blockIdx.x * 4096 % (near_bytes / 16 - 4096) + i];
    __syncthreads();
    bf16x8 a0[5], a1[5];
    bf16x8 w[3][4];
    for (int i = 0; i < 5; i++) { a0[i] = __builtin_bit_cast(bf16x8, lds[(i * 64 + lane) & 4095]); a1[i] = a0[i]; }
    f32x4 acc[5][4] = {};
    const size_t nwin = near_bytes / (gridDim.x * 8), fwin = far_bytes / (gridDim.x * 8);
    const uint64_t nbase = (uint64_t)near_buf + (size_t)(blockIdx.x * 8 + wave) * nwin;
    const uint64_t fbase = (uint64_t)far_buf + (size_t)(blockIdx.x * 8 + wave) * fwin;
    // W windows are shared by the wave pair (w, w+4)
    const uint4* wn_base = near_buf + ((size_t)(blockIdx.x * 4 + (wave & 3)) * (near_bytes / (gridDim.x * 4))) / 16 + lane;
    const uint4* wf_base = far_buf + ((size_t)(blockIdx.x * 4 + (wave & 3)) * (far_bytes / (gridDim.x * 4))) / 16 + lane;
    const unsigned wn_win = (unsigned)(near_bytes / (gridDim.x * 4) / 16), wf_win = (unsigned)(far_bytes / (gridDim.x * 4) / 16);
    unsigned noff = 0, foff = 0, rd = wave * 97, wno = 0, wfo = 0;
    uint4* sink = lds + 4096 + wave * 512;
    auto load_w = [&](bf16x8 (&dst)[4], int far) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j < far) { dst[j] = *(const bf16x8*)(wf_base + wfo); wfo += 64; if (wfo >= wf_win) wfo = 0; }
            else         { dst[j] = *(const bf16x8*)(wn_base + wno); wno += 64; if (wno >= wn_win) wno = 0; }
        }
    };
    load_w(w[0], 0); load_w(w[1], 0);
    for (int it = 0; it < iters; it += 6) {           // 6 K-steps = 12 sub-steps: the three W register sets rotate with static indices
#pragma unroll
        for (int ks = 0; ks < 12; ks++) {
            if ((ks & 1) == 0) {
#pragma unroll
                for (int g = 0; g < ((ks & 2) == 0 ? 3 : 2); g++) {
                    if ((ks & 3) == 0 && g < FAR_A) { glds16_s(fbase + foff, lane * 16, sink + g * 64); foff += 1024; if (foff >= fwin) foff = 0; }
                    else                            { glds16_s(nbase + noff, lane * 16, sink + (g + 3 * ((ks >> 1) & 1)) * 64); noff += 1024; if (noff >= nwin) noff = 0; }
                }
            }
            bf16x8* cur = (ks & 1) ? a1 : a0;
            bf16x8* nxt = (ks & 1) ? a0 : a1;
            rd = (rd + 577) & 4095;
#pragma unroll
            for (int i = 0; i < 5; i++) nxt[i] = __builtin_bit_cast(bf16x8, lds[(rd + i * 64 + lane) & 4095]);
            load_w(w[(ks + 2) % 3], (ks & 3) == 0 ? FAR_W : 0);
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(cur[i]), "v"(w[ks % 3][j]));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 5; i++) for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename K>
static void run_feed(const char* name, K kern, const uint4* nb, size_t nbytes, const uint4* fb, size_t fbytes, float* d_out, int iters,
                     int nwaves = 8, double mfma_per_kstep = 40.0) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double best = 0, sum = 0; const int reps = 5;
    for (int r = 0; r < reps + 1; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256), dim3(nwaves * 64), 0, 0, nb, nbytes, fb, fbytes, d_out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double tf = mfma_per_kstep * 16384 * iters * 256 * nwaves / (ms * 1e-3) / 1e12;
        if (r == 0) continue;
        sum += tf; if (tf > best) best = tf;
        if (r == reps) {
            unsigned long long c[4] = {0, 0, 0, 0};
            (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c));
            const double ghz = c[3] > c[1] ? (double)(c[2] - c[0]) / (double)(c[3] - c[1]) * 0.1 : 0.0;
            // cycles one SIMD spends per 40-MFMA K-step of ONE of its two waves' tiles (MFMA alone: 640 per wave)
            const double cyc = ghz * 1e9 * (ms * 1e-3) / iters;
            printf("%-44s %8.2f ms/launch  mean %7.1f TF/s  best %7.1f TF/s  clock %.2f GHz  %5.0f cycles per K-step\n", name, ms, sum / reps, best, ghz, cyc);
        }
    }
}

template <typename K>
static void run(const char* name, K kern, int frags, double flop_per_iter_per_wave, const uint4* d_rand, const uint4* d_zero, float* d_out,
                int blocks, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int data = 0; data < 2; data++) {
        const uint4* src = data ? d_zero : d_rand;
        double best = 0, sum = 0; int reps = 6;
        for (int r = 0; r < reps + 1; r++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, src, d_out, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            double tf = flop_per_iter_per_wave * iters * blocks * 4 / (ms * 1e-3) / 1e12;
            if (r == 0) continue;              // first launch warms the clocks
            sum += tf; if (tf > best) best = tf;
            if (r == reps) printf("%-22s %-6s  %8.2f ms/launch  mean %7.1f TF/s  best %7.1f TF/s\n", name, data ? "zero" : "random", ms, sum / reps, best);
        }
    }
}

int main() {
    const int blocks = 256 * 2;                 // 2 workgroups x 4 waves per CU = 2 waves per SIMD
    const int maxfrag = 16;
    size_t n = (size_t)blocks * 256 * maxfrag;
    std::vector<uint4> h(n);
    srand(1);
    uint16_t* hb = (uint16_t*)h.data();
    for (size_t i = 0; i < n * 8; i++) {
        // N(0,1)-like bf16 values: sum of 4 uniforms, centred
        float v = ((rand() & 0xffff) + (rand() & 0xffff) + (rand() & 0xffff) + (rand() & 0xffff)) / 65536.f - 2.f;
        hb[i] = bf16_bits(v * 1.7f);
    }
    uint4 *d_rand, *d_zero; float* d_out;
    CHECK(hipMalloc(&d_rand, n * 16)); CHECK(hipMalloc(&d_zero, n * 16)); CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    CHECK(hipMemcpy(d_rand, h.data(), n * 16, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_zero, 0, n * 16));
    const int iters = 200000;                   // ~50-100 ms per launch: long enough for the power manager to settle
    // 16x16x32: 2*16*16*32 = 16384 FLOP per instruction per wave; 32x32x16: 2*32*32*16 = 32768
    run("16x16x32 tile 4x4", probe16<4, 4>, 8, 16.0 * 16384, d_rand, d_zero, d_out, blocks, iters);
    run("16x16x32 tile 5x4", probe16<5, 4>, 9, 20.0 * 16384, d_rand, d_zero, d_out, blocks, iters);
    run("32x32x16 tile 2x2", probe32<2, 2>, 4, 4.0 * 32768, d_rand, d_zero, d_out, blocks, iters * 2);
    run("32x32x16 tile 3x2", probe32<3, 2>, 5, 6.0 * 32768, d_rand, d_zero, d_out, blocks, iters * 2);
    run("32x32x16 tile 4x2", probe32<4, 2>, 6, 8.0 * 32768, d_rand, d_zero, d_out, blocks, iters);

    // feeding cost: near buffer 16 MiB (8-KiB window per wave: L2 hits), far buffer 128 MiB of a 1-GiB allocation (misses L2)
    {
        const size_t nbytes = 16u << 20, fbytes = 128u << 20;      // far: 64-KiB window per wave, Infinity-Cache sized
        uint4 *nb, *fb; float* o2;
        CHECK(hipMalloc(&nb, nbytes)); CHECK(hipMalloc(&fb, (size_t)1 << 30)); CHECK(hipMalloc(&o2, 256 * 512 * 4));
        for (size_t off = 0; off < nbytes; off += n * 16 < nbytes - off ? n * 16 : nbytes - off)
            CHECK(hipMemcpy((char*)nb + off, h.data(), n * 16 < nbytes - off ? n * 16 : nbytes - off, hipMemcpyHostToDevice));
        for (size_t off = 0; off < ((size_t)1 << 30); off += nbytes) CHECK(hipMemcpy((char*)fb + off, nb, nbytes, hipMemcpyDeviceToDevice));
        const int it2 = 60000;
        run_feed("MFMA only (operands fixed)        R0 G0", feed<0, 0, 0>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("+ fragment reads                  R9 G0", feed<9, 0, 0>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("+ LDS-DMA, all L2 hits            R9 G6.5", feed<9, 13, 0>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("+ 2 of 13 pieces miss L2          R9 G6.5 far2", feed<9, 13, 2>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("  the same from a 1-GiB window (HBM) R9 G6.5 far2", feed<9, 13, 2>, nb, nbytes, fb, (size_t)1 << 30, o2, it2);
        run_feed("  4 of 13 pieces miss L2          R9 G6.5 far4", feed<9, 13, 4>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("fewer fragment reads (80x128/wave) R6 G6.5 far2", feed<6, 13, 2>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("half the DMA (tile twice as big)  R9 G3.5 far1", feed<9, 7, 1>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("both                              R6 G3.5 far1", feed<6, 7, 1>, nb, nbytes, fb, fbytes, o2, it2);
        // the same 160x256 tile on FOUR waves of 80x128 (one per SIMD): 80 MFMAs, 2 x 13 reads, 13 pieces per wave per K-step
        run_feed("4 waves x 80x128, same tile       R13/40 G13 far4/26", feed<13, 26, 4, 5, 8, 4>, nb, nbytes, fb, fbytes, o2, it2, 4, 80.0);
        // a 320x256 tile on 8 waves of 80x128: 80 MFMAs, 2 x 13 reads, 9.2 pieces per wave per K-step; fabric share 470 MB / 2.4 GB
        run_feed("320x256, one fragment set/wave    R13/40 G9 far4/18", feed<13, 18, 4, 5, 8, 8, 13, true>, nb, nbytes, fb, fbytes, o2, it2 / 2, 8, 80.0);
        run_feed("320x256, one fragment set/wave    R13/40 G9 far3/18", feed<13, 18, 3, 5, 8, 8, 13, true>, nb, nbytes, fb, fbytes, o2, it2 / 2, 8, 80.0);
        run_feed("160x256, one fragment set/wave    R9 G6.5 far2", feed<9, 13, 2, 5, 4, 8, 9, true>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("reads issued, 3 of 9 unused       R9(6 used) G6.5 far2", feed<9, 13, 2, 5, 4, 8, 6>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("W direct to registers, A by DMA   far A1/5 W3/16", feed_direct<1, 3>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("W direct to registers, all L2 hits", feed_direct<0, 0>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("4 waves x 80x128, MFMA + reads only", feed<13, 0, 0, 5, 8, 4>, nb, nbytes, fb, fbytes, o2, it2, 4, 80.0);
        // the same mixes on all-zero data: what the instruction streams sustain when power is not the limit
        CHECK(hipMemset(nb, 0, nbytes)); CHECK(hipMemset(fb, 0, (size_t)1 << 30));
        run_feed("ZEROS  MFMA + fragment reads      R9 G0", feed<9, 0, 0>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("ZEROS  + LDS-DMA, all L2 hits     R9 G6.5", feed<9, 13, 0>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("ZEROS  + 2 of 13 miss L2          R9 G6.5 far2", feed<9, 13, 2>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("ZEROS  no M0 save/restore         R9 G6.5 far2", feed<9, 13, 2, 5, 4, 8, 9, false, true>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("ZEROS  pieces by buffer_load lds  R9 G6.5 far2", feed<9, 13, 2, 5, 4, 8, 9, false, false, true>, nb, nbytes, fb, fbytes, o2, it2);
        run_feed("ZEROS  320x256 mix                R13/40 G9 far3/18", feed<13, 18, 3, 5, 8, 8, 13, true>, nb, nbytes, fb, fbytes, o2, it2 / 2, 8, 80.0);
    }
    return 0;
}
