// scan_sort.hip -- prefix sum and stable LSD radix sort for gfx950 (wave64).
//
// Replaces the reference op's cub::DeviceScan::InclusiveSum and
// cub::DeviceRadixSort::SortPairs (SURVEY 2.1 rows "scan" and "SortPairs", [UPSTREAM]).
//
// Radix pass = three launches:
//   radix_hist     per-workgroup digit histogram (LDS atomics), digit-major global layout
//   scan           exclusive scan of the 256 x nblocks counters -> global base of (digit, block)
//   radix_scatter  stable ranking: each wave owns a contiguous run of the workgroup's keys and
//                  ranks 64 keys per round with a wave64 ballot match (8 ballots for an 8-bit
//                  digit); per-wave digit counters live in LDS; waves are combined by a
//                  prefix over the 4 wave counters.  No inter-workgroup communication inside a
//                  launch, so no agent-scope fences are needed.
#include "common.h"

// ------------------------------------------------------------------------------------ scan
// 3-phase scan: block sums -> spine (one workgroup, loops) -> block scan with carry-in.
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const uint32_t* __restrict__ in, size_t n,
                                                                   uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t wsum[SCAN_THREADS / WAVE];
    size_t base = (size_t)blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        size_t idx = base + (size_t)i * SCAN_THREADS + threadIdx.x;
        if (idx < n) s += in[idx];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of block sums in place, single workgroup
__global__ __launch_bounds__(SCAN_THREADS) void scan_spine_kernel(uint32_t* __restrict__ sums, size_t nb) {
    __shared__ uint32_t wtot[SCAN_THREADS / WAVE];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t base = 0; base < nb; base += SCAN_THREADS) {
        size_t idx = base + threadIdx.x;
        uint32_t v = idx < nb ? sums[idx] : 0;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wtot[w];
        uint32_t carry = carry_s;
        if (idx < nb) sums[idx] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == SCAN_THREADS - 1) carry_s = carry + woff + inc;
        __syncthreads();
    }
}

template <bool INCLUSIVE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_final_kernel(const uint32_t* __restrict__ in,
                                                                  uint32_t* __restrict__ out, size_t n,
                                                                  const uint32_t* __restrict__ block_offs) {
    // blocked arrangement: thread t owns SCAN_ITEMS consecutive elements
    __shared__ uint32_t wtot[SCAN_THREADS / WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t tsum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        tsum += v[i];
    }
    uint32_t inc = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t run = block_offs[blockIdx.x] + inc - tsum;
    for (int w = 0; w < wave; ++w) run += wtot[w];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        uint32_t ex = run;
        run += v[i];
        if (base + i < n) out[base + i] = INCLUSIVE ? run : ex;
    }
}

void launch_exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* scratch, bool inclusive,
                               hipStream_t s) {
    if (n == 0) return;
    size_t nb = scan_blocks(n);
    scan_reduce_kernel<<<dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s>>>(in, n, scratch);
    scan_spine_kernel<<<dim3(1), dim3(SCAN_THREADS), 0, s>>>(scratch, nb);
    if (inclusive)
        scan_final_kernel<true><<<dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s>>>(in, out, n, scratch);
    else
        scan_final_kernel<false><<<dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s>>>(in, out, n, scratch);
}

// ------------------------------------------------------------------------------------ radix sort
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(const uint32_t* __restrict__ keys, size_t n,
                                                                  int shift, uint32_t* __restrict__ hist,
                                                                  unsigned nblocks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    size_t base = (size_t)blockIdx.x * SORT_TILE;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        size_t idx = base + (size_t)i * SORT_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                     const uint32_t* __restrict__ vals_in,
                                                                     uint32_t* __restrict__ keys_out,
                                                                     uint32_t* __restrict__ vals_out, size_t n,
                                                                     int shift, const uint32_t* __restrict__ offs,
                                                                     unsigned nblocks) {
    constexpr int NW = SORT_THREADS / WAVE;          // 4 waves
    constexpr int WAVE_KEYS = SORT_TILE / NW;        // 1024 consecutive keys per wave
    __shared__ uint32_t wcnt[NW][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < NW; ++w) wcnt[w][threadIdx.x] = 0;
    __syncthreads();
    const size_t wbase = (size_t)blockIdx.x * SORT_TILE + (size_t)wave * WAVE_KEYS;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS], pos[SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        size_t idx = wbase + (size_t)r * WAVE + lane;
        bool ok = idx < n;
        key[r] = ok ? keys_in[idx] : 0xFFFFFFFFu;
        val[r] = ok ? vals_in[idx] : 0u;
    }
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        size_t idx = wbase + (size_t)r * WAVE + lane;
        bool ok = idx < n;
        uint32_t d = (key[r] >> shift) & 255u;
        // lanes holding the same digit (among valid lanes)
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            uint64_t bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        uint32_t before = __popcll(peers & lt_mask);
        uint32_t cnt = __popcll(peers);
        uint32_t basep = 0;
        if (ok) basep = wcnt[wave][d];
        wave_sync();
        if (ok && before == 0) wcnt[wave][d] = basep + cnt;
        wave_sync();
        pos[r] = basep + before;
    }
    __syncthreads();
    {   // digit = threadIdx.x: turn per-wave counts into global bases
        uint32_t g = offs[(size_t)threadIdx.x * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            uint32_t c = wcnt[w][threadIdx.x];
            wcnt[w][threadIdx.x] = g;
            g += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        size_t idx = wbase + (size_t)r * WAVE + lane;
        if (idx < n) {
            uint32_t d = (key[r] >> shift) & 255u;
            uint32_t dst = wcnt[wave][d] + pos[r];
            keys_out[dst] = key[r];
            vals_out[dst] = val[r];
        }
    }
}

void launch_radix_sort_pairs(uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, size_t n, int nbits,
                             uint32_t* scratch, uint32_t** keys_out, uint32_t** vals_out, hipStream_t s) {
    uint32_t *ki = k0, *ko = k1, *vi = v0, *vo = v1;
    if (n > 0) {
        unsigned nb = (unsigned)sort_blocks(n);
        size_t hn = (size_t)nb * 256;
        uint32_t* hist = scratch;
        uint32_t* scan_scratch = scratch + hn;
        for (int shift = 0; shift < nbits; shift += 8) {
            radix_hist_kernel<<<dim3(nb), dim3(SORT_THREADS), 0, s>>>(ki, n, shift, hist, nb);
            launch_exclusive_scan_u32(hist, hist, hn, scan_scratch, false, s);
            radix_scatter_kernel<<<dim3(nb), dim3(SORT_THREADS), 0, s>>>(ki, vi, ko, vo, n, shift, hist, nb);
            uint32_t* t = ki; ki = ko; ko = t;
            t = vi; vi = vo; vo = t;
        }
    }
    *keys_out = ki;
    *vals_out = vi;
}
