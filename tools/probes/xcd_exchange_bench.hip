// Micro-benchmark behind the persistent BLSTM recurrence: what does one step of an in-launch ring cost when the
// NW workgroups of a chain sit on ONE XCD and hand a [16 x H] fp32 state to each other through that XCD's L2?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_exchange_bench tools/probes/xcd_exchange_bench.hip && /tmp/xcd_exchange_bench
//
// Each chain = NW workgroups (blockIdx % 8 = chain -> observed XCD = chain).  Step s: wait until every producer of the chain has
// published step s-1, read the whole [16 x H] row block (16-byte L1-bypassing loads), run NMFMA dependent-free v_mfma_f32_16x16x4
// instructions per wave (stand-in for the recurrent product), write the own [16 x 12] slice, publish.  Every value read is
// checked against its closed form, so a stale read is counted, not hidden.
//   mode 0: plain data stores, s_waitcnt vmcnt(0), plain flag store; consumer polls / reads with sc1 (L1-bypass, L2-served) loads.
//           Correct ONLY when producer and consumer share an L2 (same XCD).
//   mode 1: sc1 (write-through) data stores + sc1 flag store; sc1 loads.  Placement independent (guide G16, R1 write-through form).
//   mode 2: plain data stores, agent release fence, sc1 flag; sc1 loads.  Placement independent (G16 release form).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int TB = 16, UW = 12;

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float expect(int chain, int s, int row, int col) {
    return (float)((chain * 131 + s * 17 + row * 3 + col) % 1021) * 0.25f + 1.0f;
}

struct Args {
    float* data;       // [chains][T][TB][HP]
    unsigned* flags;   // [chains][64]
    unsigned* xcc;     // [grid]
    unsigned long long* cycles;   // [grid]
    unsigned* errors;  // [1]
    float* sink;
    int T, H, HP, NW, nmfma;
};

template <int MODE>
__global__ __launch_bounds__(256) void ring_kernel(Args a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chain = blockIdx.x % 8, w = blockIdx.x / 8;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        a.xcc[blockIdx.x] = id & 0xf;
    }
    float* base = a.data + (size_t)chain * a.T * TB * a.HP;
    unsigned* fl = a.flags + chain * 64;
    const rsrc_t rs = make_rsrc(base, (unsigned)((size_t)a.T * TB * a.HP * 4));
    const rsrc_t rf = make_rsrc(fl, 64 * 4);
    const int row = lane & 15, kq = (lane >> 4) * 4;
    unsigned bad = 0;
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < a.T; ++s) {
        float4 hv[5];
        if (s > 0) {
            // every wave polls for itself: lanes 0..NW-1 watch one producer each
            unsigned spins = 0;
            for (;;) {
                unsigned v = (lane < a.NW) ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rf, lane * 4, 0, 16) : 0xffffffffu;
                if (__all(v >= (unsigned)s)) break;
                if (++spins > (1u << 22)) { bad |= 0x80000000u; break; }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                int k = (wave * 5 + i) * 16 + kq;
                if (k > a.H - 4) k = a.H - 4;
                const unsigned off = (unsigned)((((size_t)(s - 1) * TB + row) * a.HP + k) * 4);
                const i32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);       // aux 16 = sc1: bypass L1, L2-served
                hv[i] = __builtin_bit_cast(float4, raw);
            }
            __builtin_amdgcn_sched_barrier(0);              // all five loads in flight before the first use
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                int k = (wave * 5 + i) * 16 + kq;
                if (k > a.H - 4) k = a.H - 4;
                bad += (hv[i].x != expect(chain, s - 1, row, k)) + (hv[i].y != expect(chain, s - 1, row, k + 1)) +
                       (hv[i].z != expect(chain, s - 1, row, k + 2)) + (hv[i].w != expect(chain, s - 1, row, k + 3));
            }
            for (int m = 0; m < a.nmfma; m += 3) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[0].x, hv[1].y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[2].x, hv[3].y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[4].x, hv[0].z, acc[2], 0, 0, 0);
            }
        }
        // own slice of step s: 16 rows x 12 columns, thread t < 192
        if (tid < TB * UW) {
            const int r = tid / UW, c = w * UW + tid % UW;
            if (c < a.H) {
                const float v = expect(chain, s, r, c) + (acc[0][0] + acc[1][1] + acc[2][2]) * 0.0f;
                const unsigned off = (unsigned)((((size_t)s * TB + r) * a.HP + c) * 4);
                if (MODE == 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs, off, 0, 16);
                else base[((size_t)s * TB + r) * a.HP + c] = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (MODE == 2) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (MODE == 0) fl[w] = (unsigned)(s + 1);
            else __hip_atomic_store(fl + w, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) a.cycles[blockIdx.x] = t1 - t0;
    if (bad) atomicAdd(a.errors, bad & 0x80000000u ? 1000000u : bad);
    if (acc[0][0] == 123.456f) a.sink[0] = acc[0][0] + acc[1][0] + acc[2][0];
}

// mode 3 / 4: the data IS the flag.  Every 16-byte granule = {3 consecutive state values, tag = step + 1}, written by ONE 16-byte store
// (mode 3: plain -- same-XCD only; mode 4: sc1 write-through -- placement independent), slots double-buffered by step parity,
// zeroed before the launch.  A consumer re-loads the granules whose tag is not the awaited one; no flag, no vmcnt drain, no barrier
// on the publish side.
struct GArgs {
    float4* gran;      // [chains][2][TB][NG]
    unsigned* xcc; unsigned long long* cycles; unsigned* errors; float* sink;
    int T, NG, NW, nmfma;
};

template <int MODE>
__global__ __launch_bounds__(256) void granule_kernel(GArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chain = blockIdx.x % 8, w = blockIdx.x / 8;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        a.xcc[blockIdx.x] = id & 0xf;
    }
    float4* base = a.gran + (size_t)chain * 2 * TB * a.NG;
    const rsrc_t rs = make_rsrc(base, (unsigned)((size_t)2 * TB * a.NG * 16));
    const int row = lane & 15, q = lane >> 4;
    const int per_wave = (a.NG + 3) / 4;                      // granules of a row handled by one wave
    unsigned bad = 0;
    unsigned long long poll_cyc = 0, poll_iters = 0;
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < a.T; ++s) {
        float4 hv[7];
        if (s > 0) {
            const float want = __uint_as_float((unsigned)s);
            unsigned got = 0, spins = 0;
            const unsigned long long p0 = __builtin_readcyclecounter();
            for (;;) {
                ++poll_iters;
                // ALL granules are re-requested every round, unconditionally: a per-granule "already have it" test makes hipcc branch
                // around each load and wait for it (7 dependent L2 round trips instead of one)
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    int g = wave * per_wave + i * 4 + q;
                    if (g > a.NG - 1) g = a.NG - 1;
                    const unsigned off = (unsigned)(((size_t)(((s - 1) & 1) * TB + row) * a.NG + g) * 16);
                    hv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
                }
                __builtin_amdgcn_sched_barrier(0);
                got = 0;
#pragma unroll
                for (int i = 0; i < 7; ++i) got |= (__float_as_uint(hv[i].w) == __float_as_uint(want)) ? (1u << i) : 0u;
                if (__all(got == 0x7fu)) break;
                if (++spins > (1u << 22)) { bad |= 0x80000000u; break; }
            }
            poll_cyc += __builtin_readcyclecounter() - p0;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                int g = wave * per_wave + i * 4 + q;
                if (g > a.NG - 1) g = a.NG - 1;
                bad += (hv[i].x != expect(chain, s - 1, row, 3 * g)) + (hv[i].y != expect(chain, s - 1, row, 3 * g + 1)) +
                       (hv[i].z != expect(chain, s - 1, row, 3 * g + 2));
            }
            for (int m = 0; m < a.nmfma; m += 3) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[0].x, hv[1].y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[2].x, hv[3].y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[4].x, hv[5].z, acc[2], 0, 0, 0);
            }
        }
        // own slice: 16 rows x 4 granules (12 units), one wave's worth of 16-byte stores
        if (tid < 64) {
            const int r = tid >> 2, g = w * 4 + (tid & 3);
            if (g < a.NG) {
                const float z = (acc[0][0] + acc[1][1] + acc[2][2]) * 0.0f;
                float4 v = make_float4(expect(chain, s, r, 3 * g) + z, expect(chain, s, r, 3 * g + 1), expect(chain, s, r, 3 * g + 2),
                                       __uint_as_float((unsigned)(s + 1)));
                const unsigned off = (unsigned)(((size_t)((s & 1) * TB + r) * a.NG + g) * 16);
                if (MODE == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs, off, 0, 16);
                else base[(size_t)((s & 1) * TB + r) * a.NG + g] = v;
            }
        }
        // a slot is rewritten two steps later; a writer can be at most one step ahead of any reader of its chain (it needs every
        // reader's granules of the step in between), so parity double-buffering is enough and no barrier is needed here
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 9 && tid == 0) { a.cycles[gridDim.x] = t1 - t0; a.cycles[gridDim.x + 1] = poll_cyc; a.cycles[gridDim.x + 2] = poll_iters; }
    if (tid == 0) a.cycles[blockIdx.x] = t1 - t0;
    if (bad) atomicAdd(a.errors, bad & 0x80000000u ? 1000000u : bad);
    if (acc[0][0] == 123.456f) a.sink[0] = acc[0][0] + acc[1][0] + acc[2][0];
}

// ping-pong: two workgroups on ONE XCD (blocks 0 and 8) bounce a counter; plain 4-byte store + sc1 polling load by one lane.
// One round = two hops.
__global__ void pingpong_kernel(unsigned* buf, unsigned long long* cycles, int rounds, int sc1_store, int sleep) {
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == 8 ? 1 : -1);
    if (me < 0 || threadIdx.x != 0) return;
    const rsrc_t rs = make_rsrc(buf, 64 * 4);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0) {
            if (sc1_store) __hip_atomic_store(buf, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else buf[0] = (unsigned)r;
            while ((unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, 32 * 4, 0, 16) != (unsigned)r) { if (sleep) __builtin_amdgcn_s_sleep(1); }
        } else {
            while ((unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 16) != (unsigned)r) { if (sleep) __builtin_amdgcn_s_sleep(1); }
            if (sc1_store) __hip_atomic_store(buf + 32, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else buf[32] = (unsigned)r;
        }
    }
    cycles[me] = __builtin_readcyclecounter() - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>
void run(int NW, int nmfma, int T, int reps) {
    const int H = 300, HP = 304, chains = 8, grid = chains * NW;
    Args a{};
    a.T = T; a.H = H; a.HP = HP; a.NW = NW; a.nmfma = nmfma;
    const size_t nd = (size_t)chains * T * TB * HP;
    CK(hipMalloc(&a.data, nd * 4));
    CK(hipMalloc(&a.flags, chains * 64 * 4));
    CK(hipMalloc(&a.xcc, grid * 4));
    CK(hipMalloc(&a.cycles, grid * 8));
    CK(hipMalloc(&a.errors, 4));
    CK(hipMalloc(&a.sink, 4));
    CK(hipMemset(a.errors, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(a.data, 0, nd * 4, 0));
        CK(hipMemsetAsync(a.flags, 0, chains * 64 * 4, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(ring_kernel<MODE>, dim3(grid), dim3(256), 0, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) { best = ms < best ? ms : best; sum += ms; }
    }
    std::vector<unsigned> xcc(grid);
    std::vector<unsigned long long> cyc(grid);
    unsigned err;
    CK(hipMemcpy(xcc.data(), a.xcc, grid * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cyc.data(), a.cycles, grid * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&err, a.errors, 4, hipMemcpyDeviceToHost));
    int placed = 0;
    for (int b = 0; b < grid; ++b) placed += (xcc[b] == (unsigned)(b % 8));
    unsigned long long cmax = 0;
    for (auto c : cyc) cmax = c > cmax ? c : cmax;
    printf("mode %d NW %2d nmfma %3d T %d: %.3f us/step (best of %d; mean %.3f), %llu shader cycles/step, stale/timeouts %u (over all reps), "
           "blocks on XCD b%%8: %d/%d\n", MODE, NW, nmfma, T, best * 1e3f / T, reps - 1, sum / (reps - 1) * 1e3f / T, cmax / T, err, placed, grid);
    CK(hipFree(a.data)); CK(hipFree(a.flags)); CK(hipFree(a.xcc)); CK(hipFree(a.cycles)); CK(hipFree(a.errors)); CK(hipFree(a.sink));
}

template <int MODE>
void run_gran(int NW, int nmfma, int T, int reps) {
    const int NG = 100, chains = 8, grid = chains * NW;
    GArgs a{};
    a.T = T; a.NG = NG; a.NW = NW; a.nmfma = nmfma;
    const size_t nd = (size_t)chains * 2 * TB * NG;
    CK(hipMalloc(&a.gran, nd * 16));
    CK(hipMalloc(&a.xcc, grid * 4));
    CK(hipMalloc(&a.cycles, (grid + 4) * 8));
    CK(hipMalloc(&a.errors, 4));
    CK(hipMalloc(&a.sink, 4));
    CK(hipMemset(a.errors, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(a.gran, 0, nd * 16, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(granule_kernel<MODE>, dim3(grid), dim3(256), 0, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) { best = ms < best ? ms : best; sum += ms; }
    }
    std::vector<unsigned> xcc(grid);
    std::vector<unsigned long long> cyc(grid + 4);
    unsigned err;
    CK(hipMemcpy(xcc.data(), a.xcc, grid * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cyc.data(), a.cycles, (grid + 4) * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&err, a.errors, 4, hipMemcpyDeviceToHost));
    int placed = 0;
    for (int b = 0; b < grid; ++b) placed += (xcc[b] == (unsigned)(b % 8));
    unsigned long long cmax = 0;
    for (int b = 0; b < grid; ++b) cmax = cyc[b] > cmax ? cyc[b] : cmax;
    printf("mode %d NW %2d nmfma %3d T %d: %.3f us/step (best of %d; mean %.3f), %llu shader cycles/step, stale/timeouts %u (over all reps), "
           "blocks on XCD b%%8: %d/%d; block 9: %llu cyc/step of which polling %llu in %.2f rounds\n", MODE, NW, nmfma, T, best * 1e3f / T, reps - 1,
           sum / (reps - 1) * 1e3f / T, cmax / T, err, placed, grid, cyc[grid] / T, cyc[grid + 1] / T, (double)cyc[grid + 2] / T);
    CK(hipFree(a.gran)); CK(hipFree(a.xcc)); CK(hipFree(a.cycles)); CK(hipFree(a.errors)); CK(hipFree(a.sink));
}

void run_pingpong(int sc1_store, int sleep) {
    unsigned* buf; unsigned long long* cyc;
    CK(hipMalloc(&buf, 64 * 4)); CK(hipMalloc(&cyc, 16));
    CK(hipMemset(buf, 0, 64 * 4));
    const int rounds = 2000;
    hipLaunchKernelGGL(pingpong_kernel, dim3(16), dim3(64), 0, 0, buf, cyc, rounds, sc1_store, sleep);
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
    printf("ping-pong same XCD, %s store, sleep %d: %.0f shader cycles per hop\n", sc1_store ? "sc1" : "plain", sleep, (double)h[0] / rounds / 2);
    CK(hipFree(buf)); CK(hipFree(cyc));
}

int main() {
    const int T = 80, reps = 6;
    run_pingpong(0, 0); run_pingpong(0, 1); run_pingpong(1, 0); run_pingpong(1, 1);
    for (int nm : {0, 57}) {
        const int NW = 25;
        run<0>(NW, nm, T, reps);
        run<1>(NW, nm, T, reps);
        run<2>(NW, nm, T, reps);
        run_gran<3>(NW, nm, T, reps);
        run_gran<4>(NW, nm, T, reps);
    }
    return 0;
}
