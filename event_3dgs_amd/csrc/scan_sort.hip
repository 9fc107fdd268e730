// scan_sort.hip -- prefix sum and stable LSD radix sort for gfx950 (wave64).
//
// Replaces the reference op's cub::DeviceScan::InclusiveSum and
// cub::DeviceRadixSort::SortPairs (SURVEY 2.1 rows "scan" and "SortPairs", [UPSTREAM]).
//
// Radix pass = three launches:
//   radix_hist     per-workgroup digit histogram (LDS atomics), digit-major global layout
//   scan           exclusive scan of the 256 x nblocks counters -> global base of (digit, block); ONE launch
//                  (chained scan with decoupled look-back)
//   radix_scatter  stable ranking: each wave owns a contiguous run of the workgroup's keys and
//                  ranks 64 keys per round with a wave64 ballot match (8 ballots for an 8-bit
//                  digit); per-wave digit counters live in LDS; waves are combined by a
//                  prefix over the 4 wave counters; keys are staged through LDS so that the global stores of a
//                  store instruction cover consecutive addresses inside each digit segment.
//
// The kernels are templated on the key type: the tile sort of a frame with <= 65536 tiles moves 16-bit keys
// (a third of its bytes less per pass), the depth sort 32-bit keys.  Two extras serve the rasteriser:
//   DROP   (first pass of the depth sort) keys equal to 0xFFFFFFFF -- splats the projection culled, a third of the
//          keys of the benchmark scene -- are not counted and not scattered: the sort compacts while it sorts, the
//          kept count goes to device memory, and the later passes (and everything behind the sort) read their element
//          count from there.
//   NOKEYS (last pass of the tile sort) only the payload is written: nobody reads the sorted keys once the tile
//          ranges are known -- they are derived in the same kernel (see radix_scatter_kernel).
#include "common.h"
#include <stdlib.h>

int e3_fail(hipError_t e, const char* what);
#define LAUNCH_OK(name)                                       \
    do {                                                      \
        hipError_t _e = hipGetLastError();                    \
        if (_e != hipSuccess) return e3_fail(_e, name);       \
    } while (0)

// ------------------------------------------------------------------------------------ scan
// 3-phase scan: block sums -> spine (one workgroup, loops) -> block scan with carry-in.  Needs no zeroed state:
// used where no kernel in front of the scan could prepare the chained scan's descriptors (densify.hip).
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

int launch_exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* scratch, bool inclusive,
                              hipStream_t s) {
    if (n == 0) return 0;
    size_t nb = scan_blocks(n);
    scan_reduce_kernel<<<dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s>>>(in, n, scratch);
    LAUNCH_OK("scan_reduce_kernel");
    scan_spine_kernel<<<dim3(1), dim3(SCAN_THREADS), 0, s>>>(scratch, nb);
    LAUNCH_OK("scan_spine_kernel");
    if (inclusive)
        scan_final_kernel<true><<<dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s>>>(in, out, n, scratch);
    else
        scan_final_kernel<false><<<dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s>>>(in, out, n, scratch);
    LAUNCH_OK("scan_final_kernel");
    return 0;
}

// ------------------------------------------------------------------------------------ single-launch scan
// The same scan in ONE launch (chained scan with decoupled look-back): for the scans that sit between two kernels of
// the list-building chain, where three launches of ~5 us each cost more than the scan itself.  Protocol
// (cdna_hip_programming.md G16, form R2): one 64-bit descriptor {2-bit status, 62-bit sum} per workgroup -- the data is
// its own flag -- written and read with relaxed agent-scope atomics; logical workgroup ids come from an atomic ticket,
// so a workgroup only ever waits for workgroups that have already started.  Wave 0 of a workgroup looks back 64
// predecessors at a time.  `desc` = nb + 1 words (descriptors, then the ticket) that MUST be zero on entry: the kernel
// in front of the scan in the stream zeroes them (radix_hist_kernel / preprocess_kernel), which costs no extra command.
// The 32-bit elements may sum up to anything a uint32 holds (the 62-bit field cannot overflow; round 2 packed
// {status, 30-bit sum} into one 32-bit word, which silently corrupted the status bits beyond 2^30 instances).
constexpr unsigned long long SC_AGG = 1ull << 62, SC_PREFIX = 2ull << 62, SC_MASK = (1ull << 62) - 1ull;
template <bool INCLUSIVE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_chained_kernel(const uint32_t* in, uint32_t* out, size_t n,
                                                                    unsigned long long* desc, unsigned nb,
                                                                    volatile int* total_host, uint32_t* total_dev) {
    __shared__ uint32_t wtot[SCAN_THREADS / WAVE];
    __shared__ uint32_t s_bid, s_excl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_bid = (uint32_t)atomicAdd(desc + nb, 1ull);
    __syncthreads();
    const uint32_t bid = s_bid;
    const size_t base = (size_t)bid * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t tsum = 0;
    // a thread's 16 consecutive elements as four 16-byte loads when the tile is full and the pointer aligned
    const bool full = base + SCAN_ITEMS <= n;
    if (full && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {
        const uint4* __restrict__ in4 = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS / 4; ++i) {
            const uint4 q = in4[i];
            v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) v[i] = (base + i < n) ? in[base + i] : 0;
    }
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) tsum += v[i];
    uint32_t inc = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        const uint32_t total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        if (lane == 0)
            __hip_atomic_store(desc + bid, (bid == 0 ? SC_PREFIX : SC_AGG) | (unsigned long long)total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        int p = (int)bid - 1;                       // lane l looks at workgroup p - l
        while (p >= 0) {
            const int idx = p - lane;
            const unsigned long long d = idx >= 0 ? __hip_atomic_load(desc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                  : SC_PREFIX;             // virtual workgroup -1: prefix 0
            const uint32_t st = (uint32_t)(d >> 62);
            const unsigned long long ready = __ballot(st != 0u), pre = __ballot(st == 2u);
            const int lead = (~ready) ? __builtin_ctzll(~ready) : 64;       // published entries in a row from lane 0
            const int firstpre = pre ? __builtin_ctzll(pre) : 64;
            const int use = firstpre < lead ? firstpre + 1 : lead;          // consumed this round
            uint32_t c = lane < use ? (uint32_t)(d & SC_MASK) : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
            excl += c;
            if (firstpre < lead) break;                                     // reached an inclusive prefix
            p -= use;
            if (use == 0) __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) {
            if (bid != 0)
                __hip_atomic_store(desc + bid, SC_PREFIX | (unsigned long long)(excl + total), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
        }
    }
    __syncthreads();
    uint32_t run = s_excl + inc - tsum;
    for (int w = 0; w < wave; ++w) run += wtot[w];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        uint32_t ex = run;
        run += v[i];
        v[i] = INCLUSIVE ? run : ex;
    }
    if (full && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        uint4* __restrict__ out4 = reinterpret_cast<uint4*>(out + base);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS / 4; ++i) out4[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i)
            if (base + i < n) out[base + i] = v[i];
    }
    // the grand total, by the thread that owns the last element: to device memory (the kernels behind a compacting sort
    // pass read their element count there) and / or straight into host-visible (pinned, mapped) memory that the host
    // polls (E3_FLAG_COUNT_MAPPED) -- no separate publishing kernel, no copy command
    if (base < n && n <= base + SCAN_ITEMS) {
        if (total_dev) *total_dev = run;
        if (total_host) {
            *total_host = (int)run;
            __threadfence_system();
        }
    }
}

int launch_scan_chained_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* desc_zeroed, bool inclusive,
                            hipStream_t s, int* total_host, uint32_t* total_dev) {
    if (n == 0) return 0;
    const unsigned nb = (unsigned)scan_blocks(n);
    unsigned long long* desc = reinterpret_cast<unsigned long long*>(desc_zeroed);
    if (inclusive)
        scan_chained_kernel<true><<<dim3(nb), dim3(SCAN_THREADS), 0, s>>>(in, out, n, desc, nb, total_host, total_dev);
    else
        scan_chained_kernel<false><<<dim3(nb), dim3(SCAN_THREADS), 0, s>>>(in, out, n, desc, nb, total_host, total_dev);
    LAUNCH_OK("scan_chained_kernel");
    return 0;
}

// ------------------------------------------------------------------------------------ radix sort
// n: number of keys -- the host's figure, or (n_dev != NULL) whatever the device word holds (the kept count of a
// compacting first pass; the instance count of a forward whose buffers were sized before the count was known).  Grids
// are sized by the host's figure, which is then the CAPACITY of the buffers: workgroups behind the device count only
// contribute zero counters / exit, and a device count beyond the capacity sorts nothing at all (the caller notices the
// overflow when the count reaches it and repeats the call with larger buffers).
__device__ __forceinline__ size_t device_count(const uint32_t* __restrict__ n_dev, size_t n_host) {
    if (!n_dev) return n_host;
    const size_t n = (size_t)*n_dev;
    return n > n_host ? 0 : n;
}
template <typename KeyT, bool DROP>
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(const KeyT* __restrict__ keys, size_t n_host,
                                                                  const uint32_t* __restrict__ n_dev, int shift,
                                                                  uint32_t* __restrict__ hist, unsigned nblocks,
                                                                  unsigned long long* __restrict__ scan_desc,
                                                                  unsigned ndesc) {
    __shared__ uint32_t h[256];
    // (descriptors + ticket of the chained scan that follows in the stream: zeroed here instead of by a memset command)
    if (blockIdx.x == 0) for (unsigned t = threadIdx.x; t < ndesc; t += SORT_THREADS) scan_desc[t] = 0ull;
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t n = device_count(n_dev, n_host);
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
    constexpr int PER = 16 / (int)sizeof(KeyT);          // keys per 16-byte load
    if (base + SORT_TILE <= n && (reinterpret_cast<uintptr_t>(keys) & 15) == 0) {
        // full tile: the order inside a histogram tile does not matter, so every thread takes 16 consecutive keys as
        // 16-byte loads (all issued before the first use)
        uint4 q[SORT_ITEMS / PER];
        const uint4* __restrict__ k4 = reinterpret_cast<const uint4*>(keys + base);
#pragma unroll
        for (int i = 0; i < SORT_ITEMS / PER; ++i) q[i] = k4[(size_t)i * SORT_THREADS + threadIdx.x];
#pragma unroll
        for (int i = 0; i < SORT_ITEMS / PER; ++i) {
            const uint32_t w[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (sizeof(KeyT) == 4) {
                    if (!DROP || w[j] != 0xFFFFFFFFu) atomicAdd(&h[(w[j] >> shift) & 255u], 1u);
                } else {
                    atomicAdd(&h[((w[j] & 0xFFFFu) >> shift) & 255u], 1u);
                    atomicAdd(&h[((w[j] >> 16) >> shift) & 255u], 1u);
                }
            }
        }
    } else if (base < n) {
        // (all loads first, unconditional with a clamped index: a load inside `if (idx < n)` next to its use waits for its
        // own round trip in every unrolled iteration)
        uint32_t k[SORT_ITEMS];
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            size_t idx = base + (size_t)i * SORT_THREADS + threadIdx.x;
            k[i] = (uint32_t)keys[idx < n ? idx : n - 1];
        }
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            size_t idx = base + (size_t)i * SORT_THREADS + threadIdx.x;
            if (idx < n && (!DROP || k[i] != 0xFFFFFFFFu)) atomicAdd(&h[(k[i] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// After the stable tile sort: [start, end) of each tile.  (Used when the ranges are not derived inside the last
// scatter pass: single-pass and three-pass tile sorts.)
template <typename KeyT>
__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint32_t* __restrict__ n_dev, uint32_t n_host,
                                                          const KeyT* __restrict__ keys, uint2* __restrict__ ranges,
                                                          uint32_t nranges) {
    const uint32_t I = (uint32_t)device_count(n_dev, n_host);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    const uint32_t t = keys[i];
    if (t >= nranges) return;
    if (i == 0 || (uint32_t)keys[i - 1] != t) ranges[t].x = i;
    if (i == I - 1 || (uint32_t)keys[i + 1] != t) ranges[t].y = i + 1;
}

// LDS of one scatter workgroup: per-wave digit counters, the staged (key, value) pairs, per-digit global offsets.
template <typename KeyT>
struct ScatterLds {
    uint32_t wcnt[SORT_THREADS / WAVE][256];
    KeyT key[SORT_TILE];
    uint32_t val[SORT_TILE];
    uint32_t glob[256];     // global position minus local position, per digit
    uint32_t wtot[SORT_THREADS / WAVE];
    uint32_t nvalid;
    uint32_t seg_tmp[8];    // seg_block()
};

// RANGES (last pass of a TWO-pass tile sort, NOKEYS): the input is sorted by the low digit and the workgroups of this
// pass never straddle two low digits -- workgroup b handles (at most) SORT_TILE keys of ONE low-digit segment
// (`seg` = the 257 segment boundaries the first pass left behind).  Then, in the digit-major scanned histogram,
// offs[h][first workgroup of low digit l] is exactly where the run of tile (h << 8 | l) starts, and the run ends
// where the last workgroup of l puts its last key of h: the tile ranges fall out of numbers this kernel holds anyway --
// no pass over the sorted keys (tile_ranges_kernel), no sorted keys at all.
struct SegBlock { uint32_t first, count, low, is_first, is_last, first_block, valid; };
__device__ __forceinline__ SegBlock seg_block(const uint32_t* __restrict__ seg, uint32_t b, uint32_t* s_tmp /* >= 8 words */) {
    // thread d: segment d = [seg[d], seg[d+1]) is cut into max(1, ceil(len / SORT_TILE)) workgroups -- an EMPTY segment keeps
    // one (keyless) workgroup, so that every low digit, hence every key value, has a workgroup that writes its ranges --;
    // exclusive scan over d.  (sum <= n / SORT_TILE + 256 workgroups: seg_blocks())
    const int d = threadIdx.x, lane = d & 63, wave = d >> 6;
    const uint32_t s0 = seg[d], s1 = seg[d + 1];
    const uint32_t nblk_keys = (s1 - s0 + SORT_TILE - 1) / SORT_TILE;
    const uint32_t nblk = nblk_keys ? nblk_keys : 1u;
    uint32_t inc = nblk;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    uint32_t bf = inc - nblk;
    for (int w = 0; w < wave; ++w) bf += s_tmp[w];
    __syncthreads();
    if (d == 0) s_tmp[4] = 0xFFFFFFFFu;
    __syncthreads();
    if (b >= bf && b < bf + nblk) {
        const uint32_t k = b - bf;
        s_tmp[0] = s0 + k * SORT_TILE;
        s_tmp[1] = (s1 - s0 - k * SORT_TILE) < (uint32_t)SORT_TILE ? (s1 - s0 - k * SORT_TILE) : (uint32_t)SORT_TILE;
        s_tmp[2] = (uint32_t)d;
        s_tmp[3] = (k == 0 ? 1u : 0u) | (k + 1 == nblk ? 2u : 0u);
        s_tmp[4] = 0u;
        s_tmp[5] = bf;
    }
    __syncthreads();
    SegBlock r;
    r.valid = s_tmp[4] ? 0u : 1u;
    r.first = s_tmp[0]; r.count = r.valid ? s_tmp[1] : 0u; r.low = s_tmp[2];
    r.is_first = r.valid ? (s_tmp[3] & 1u) : 0u; r.is_last = r.valid ? (s_tmp[3] >> 1) : 0u;
    r.first_block = s_tmp[5];
    __syncthreads();
    return r;
}

// same decomposition for the histogram of that pass
template <typename KeyT>
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_seg_kernel(const KeyT* __restrict__ keys,
                                                                      const uint32_t* __restrict__ seg, int shift,
                                                                      uint32_t* __restrict__ hist, unsigned nblocks,
                                                                      unsigned long long* __restrict__ scan_desc,
                                                                      unsigned ndesc) {
    __shared__ uint32_t h[256];
    __shared__ uint32_t s_tmp[8];
    if (blockIdx.x == 0) for (unsigned t = threadIdx.x; t < ndesc; t += SORT_THREADS) scan_desc[t] = 0ull;
    h[threadIdx.x] = 0;
    const SegBlock sb = seg_block(seg, blockIdx.x, s_tmp);      // (contains the barriers that publish h = 0)
    if (sb.count) {
        uint32_t k[SORT_ITEMS];
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            const uint32_t r = (uint32_t)i * SORT_THREADS + threadIdx.x;
            k[i] = (uint32_t)keys[(size_t)sb.first + (r < sb.count ? r : sb.count - 1)];
        }
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            const uint32_t r = (uint32_t)i * SORT_THREADS + threadIdx.x;
            if (r < sb.count) atomicAdd(&h[(k[i] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

template <typename KeyT, bool DROP, bool NOKEYS, bool RANGES>
__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(const KeyT* __restrict__ keys_in,
                                                                     const uint32_t* __restrict__ vals_in,
                                                                     KeyT* __restrict__ keys_out,
                                                                     uint32_t* __restrict__ vals_out, size_t n_host,
                                                                     const uint32_t* __restrict__ n_dev, int shift,
                                                                     const uint32_t* __restrict__ offs,
                                                                     unsigned nblocks,
                                                                     uint32_t* __restrict__ seg /* RANGES: in; else: out (or null) */,
                                                                     uint2* __restrict__ ranges, uint32_t nranges,
                                                                     uint32_t* __restrict__ lpt_cnt,
                                                                     uint32_t* __restrict__ lpt_list) {
    constexpr int NW = SORT_THREADS / WAVE;          // 4 waves
    constexpr int WAVE_KEYS = SORT_TILE / NW;        // 1024 consecutive keys per wave
    __shared__ ScatterLds<KeyT> L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t n = device_count(n_dev, n_host);
    size_t block_first = (size_t)blockIdx.x * SORT_TILE;
    SegBlock sb;
    if (RANGES) {
        sb = seg_block(seg, blockIdx.x, L.seg_tmp);
        block_first = sb.first;
        n = (size_t)sb.first + sb.count;
        if (!sb.valid) return;                        // (uniform; no workgroup barrier is pending.  A valid workgroup
                                                      // without keys -- an empty segment -- still writes its ranges below)
    } else {
        // segment boundaries of THIS pass's digit for a later RANGES pass: position of the first key of digit d = the
        // scanned counter of (d, workgroup 0)
        if (seg && blockIdx.x == 0) {
            seg[threadIdx.x] = offs[(size_t)threadIdx.x * nblocks];
            if (threadIdx.x == 0) seg[256] = (uint32_t)n;
        }
        if (block_first >= n) return;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) L.wcnt[w][threadIdx.x] = 0;
    __syncthreads();
    const size_t wbase = block_first + (size_t)wave * WAVE_KEYS;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS], pos[SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        size_t idx = wbase + (size_t)r * WAVE + lane;
        bool ok = idx < n;
        key[r] = ok ? (uint32_t)keys_in[idx] : 0xFFFFFFFFu;
        val[r] = ok ? (vals_in ? vals_in[idx] : (uint32_t)idx) : 0u;   // no payload array: the payload is the index
    }
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        size_t idx = wbase + (size_t)r * WAVE + lane;
        const bool ok = idx < n && (!DROP || key[r] != 0xFFFFFFFFu);
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
        if (ok) basep = L.wcnt[wave][d];
        wave_sync();
        if (ok && before == 0) L.wcnt[wave][d] = basep + cnt;
        wave_sync();
        pos[r] = ok ? basep + before : 0xFFFFFFFFu;
    }
    __syncthreads();
    // ---- digit = threadIdx.x: offs holds the global position of this workgroup's first key of each digit
    const int d = threadIdx.x;
    const uint32_t gfirst = offs[(size_t)d * nblocks + blockIdx.x];
    uint32_t c[NW], total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { c[w] = L.wcnt[w][d]; total += c[w]; }
    // workgroup-local exclusive scan of the digit totals
    uint32_t inc = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) L.wtot[wave] = inc;
    __syncthreads();
    uint32_t lbase = inc - total;
    for (int w = 0; w < wave; ++w) lbase += L.wtot[w];
    L.glob[d] = gfirst - lbase;
    if (d == SORT_THREADS - 1) L.nvalid = lbase + total;
    if (RANGES && sb.is_last) {
        // runs of the tiles (d << shift | low): they start where the FIRST workgroup of the low digit puts its first key
        // of d (its scanned counter) and end behind the last key of d of the LAST one -- this workgroup, which therefore
        // knows both ends: one plain 8-byte store per tile, every tile written exactly once (every low digit has a last
        // workgroup, seg_block), tiles without a key as the documented (0, 0).
        const uint32_t tile = ((uint32_t)d << shift) | sb.low;
        const bool mine = tile < nranges;             // (digits beyond the last tile id hold no key)
        const uint32_t start = offs[(size_t)d * nblocks + sb.first_block], end = gfirst + total;
        if (mine) ranges[tile] = end > start ? make_uint2(start, end) : make_uint2(0u, 0u);
        if (lpt_cnt) {
            // launch order of the compositing kernel (ImageState::lpt_*): the tile joins the list of its cost class.  One
            // atomic per (wave, class): the lanes of a class are matched with ballots, the first one reserves for all.
            const uint32_t cls = e3_lpt_class(end - start);
            unsigned long long peers = __ballot(mine);
#pragma unroll
            for (int b = 0; b < 7; ++b) {               // (E3_LPT_CLASSES = 128)
                const unsigned long long bal = __ballot((cls >> b) & 1u);
                peers &= ((cls >> b) & 1u) ? bal : ~bal;
            }
            if (mine) {
                const int leader = __builtin_ctzll(peers);
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(&lpt_cnt[cls], (uint32_t)__popcll(peers));
                base = (uint32_t)__shfl((int)base, leader, 64);
                lpt_list[(size_t)cls * nranges + base + (uint32_t)__popcll(peers & ((1ull << lane) - 1ull))] = tile;
            }
        }
    }
    uint32_t l = lbase;
#pragma unroll
    for (int w = 0; w < NW; ++w) { L.wcnt[w][d] = l; l += c[w]; }
    __syncthreads();
    // Keys are first placed at their workgroup-local sorted position in LDS and then streamed out by consecutive
    // threads, so the lanes of a store instruction write consecutive addresses inside each digit's segment (avg 16
    // keys) instead of 4-key fragments per (wave, digit).
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        if (pos[r] != 0xFFFFFFFFu) {
            uint32_t dg = (key[r] >> shift) & 255u;
            uint32_t lp = L.wcnt[wave][dg] + pos[r];
            L.key[lp] = (KeyT)key[r];
            L.val[lp] = val[r];
        }
    }
    __syncthreads();
    const uint32_t nvalid = L.nvalid;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        uint32_t lp = (uint32_t)j * SORT_THREADS + threadIdx.x;
        if (lp < nvalid) {
            uint32_t k = (uint32_t)L.key[lp];
            uint32_t dst = L.glob[(k >> shift) & 255u] + lp;
            if (!NOKEYS) keys_out[dst] = (KeyT)k;
            vals_out[dst] = L.val[lp];
        }
    }
}

// ------------------------------------------------------------------------------------ wide-digit passes (depth sort)
// The depth sort's keys are the 32 bits of a positive float: three passes of 11 + 11 + 10 bits instead of four of 8
// (2048-bin histograms: 8 KB of LDS; the review of round 5 asked for this build).  Same three launches per pass and the
// same ranking scheme as the 8-bit kernels above -- 11 ballots per round instead of 8, 16-bit per-wave counters (a wave
// ranks 1024 keys, a workgroup 4096), eight consecutive digits per thread in the digit phase.  DROP / identity payload in
// the first pass and NOKEYS in the last one as above.
constexpr int WIDE_BITS = 11;
constexpr int WIDE_BINS = 1 << WIDE_BITS;
constexpr int WIDE_DPT = WIDE_BINS / SORT_THREADS;      // digits per thread in the digit phase (8)

template <bool DROP>
__global__ __launch_bounds__(SORT_THREADS) void wide_hist_kernel(const uint32_t* __restrict__ keys, size_t n_host,
                                                                 const uint32_t* __restrict__ n_dev, int shift,
                                                                 uint32_t* __restrict__ hist, unsigned nblocks,
                                                                 unsigned long long* __restrict__ scan_desc,
                                                                 unsigned ndesc) {
    __shared__ uint32_t h[WIDE_BINS];
    if (blockIdx.x == 0) for (unsigned t = threadIdx.x; t < ndesc; t += SORT_THREADS) scan_desc[t] = 0ull;
#pragma unroll
    for (int j = 0; j < WIDE_DPT; ++j) h[j * SORT_THREADS + threadIdx.x] = 0;
    __syncthreads();
    const size_t n = device_count(n_dev, n_host);
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
    if (base + SORT_TILE <= n && (reinterpret_cast<uintptr_t>(keys) & 15) == 0) {
        uint4 q[SORT_ITEMS / 4];
        const uint4* __restrict__ k4 = reinterpret_cast<const uint4*>(keys + base);
#pragma unroll
        for (int i = 0; i < SORT_ITEMS / 4; ++i) q[i] = k4[(size_t)i * SORT_THREADS + threadIdx.x];
#pragma unroll
        for (int i = 0; i < SORT_ITEMS / 4; ++i) {
            const uint32_t w[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!DROP || w[j] != 0xFFFFFFFFu) atomicAdd(&h[(w[j] >> shift) & (WIDE_BINS - 1)], 1u);
        }
    } else if (base < n) {
        uint32_t k[SORT_ITEMS];
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            size_t idx = base + (size_t)i * SORT_THREADS + threadIdx.x;
            k[i] = keys[idx < n ? idx : n - 1];
        }
#pragma unroll
        for (int i = 0; i < SORT_ITEMS; ++i) {
            size_t idx = base + (size_t)i * SORT_THREADS + threadIdx.x;
            if (idx < n && (!DROP || k[i] != 0xFFFFFFFFu)) atomicAdd(&h[(k[i] >> shift) & (WIDE_BINS - 1)], 1u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WIDE_DPT; ++j) {
        const int d = j * SORT_THREADS + threadIdx.x;
        hist[(size_t)d * nblocks + blockIdx.x] = h[d];
    }
}

struct WideLds {
    union {
        uint16_t wcnt[SORT_THREADS / WAVE][WIDE_BINS];   // per-wave digit counters, then workgroup-local offsets (< 4096)
        uint32_t glob[WIDE_BINS];                        // (once the keys are staged) global minus local position, per digit
    };
    uint32_t key[SORT_TILE];
    uint32_t val[SORT_TILE];
    uint32_t wtot[SORT_THREADS / WAVE];
    uint32_t nvalid;
};                                                       // 48 KB: three workgroups per CU
static_assert(SORT_TILE <= 65535, "16-bit workgroup-local offsets");

template <bool DROP, bool NOKEYS>
__global__ __launch_bounds__(SORT_THREADS) void wide_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                    const uint32_t* __restrict__ vals_in,
                                                                    uint32_t* __restrict__ keys_out,
                                                                    uint32_t* __restrict__ vals_out, size_t n_host,
                                                                    const uint32_t* __restrict__ n_dev, int shift,
                                                                    const uint32_t* __restrict__ offs, unsigned nblocks) {
    constexpr int NW = SORT_THREADS / WAVE;
    constexpr int WAVE_KEYS = SORT_TILE / NW;
    __shared__ WideLds L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t n = device_count(n_dev, n_host);
    const size_t block_first = (size_t)blockIdx.x * SORT_TILE;
    if (block_first >= n) return;
    // (32-bit stores: two counters each)
    {
        uint32_t* z = reinterpret_cast<uint32_t*>(&L.wcnt[0][0]);
#pragma unroll
        for (int j = 0; j < NW * WIDE_BINS / 2 / SORT_THREADS; ++j) z[j * SORT_THREADS + threadIdx.x] = 0u;
    }
    // global positions of this workgroup's first key of each digit (digit-major scanned histogram): issued early
    uint32_t gfirst[WIDE_DPT];
#pragma unroll
    for (int j = 0; j < WIDE_DPT; ++j) gfirst[j] = offs[(size_t)(threadIdx.x * WIDE_DPT + j) * nblocks + blockIdx.x];
    __syncthreads();
    const size_t wbase = block_first + (size_t)wave * WAVE_KEYS;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t key[SORT_ITEMS], val[SORT_ITEMS], pos[SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        size_t idx = wbase + (size_t)r * WAVE + lane;
        bool ok = idx < n;
        key[r] = ok ? keys_in[idx] : 0xFFFFFFFFu;
        val[r] = ok ? (vals_in ? vals_in[idx] : (uint32_t)idx) : 0u;
    }
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        size_t idx = wbase + (size_t)r * WAVE + lane;
        const bool ok = idx < n && (!DROP || key[r] != 0xFFFFFFFFu);
        const uint32_t d = (key[r] >> shift) & (WIDE_BINS - 1);
        uint64_t peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < WIDE_BITS; ++b) {
            uint64_t bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const uint32_t before = __popcll(peers & lt_mask);
        const uint32_t cnt = __popcll(peers);
        uint32_t basep = 0;
        if (ok) basep = L.wcnt[wave][d];
        wave_sync();
        if (ok && before == 0) L.wcnt[wave][d] = (uint16_t)(basep + cnt);
        wave_sync();
        pos[r] = ok ? basep + before : 0xFFFFFFFFu;
    }
    __syncthreads();
    // ---- digit phase: thread t owns the WIDE_DPT consecutive digits t * WIDE_DPT ...
    uint32_t c[WIDE_DPT][NW], tot[WIDE_DPT], tsum = 0;
#pragma unroll
    for (int j = 0; j < WIDE_DPT; ++j) {
        tot[j] = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { c[j][w] = L.wcnt[w][threadIdx.x * WIDE_DPT + j]; tot[j] += c[j][w]; }
        tsum += tot[j];
    }
    uint32_t inc = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) L.wtot[wave] = inc;
    __syncthreads();
    uint32_t lbase = inc - tsum;
    for (int w = 0; w < wave; ++w) lbase += L.wtot[w];
#pragma unroll
    for (int j = 0; j < WIDE_DPT; ++j) {
        const int d = threadIdx.x * WIDE_DPT + j;
        gfirst[j] -= lbase;                       // global minus local position of digit d (goes to L.glob below)
        uint32_t l = lbase;
#pragma unroll
        for (int w = 0; w < NW; ++w) { L.wcnt[w][d] = (uint16_t)l; l += c[j][w]; }
        lbase += tot[j];
    }
    if (threadIdx.x == SORT_THREADS - 1) L.nvalid = lbase;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        if (pos[r] != 0xFFFFFFFFu) {
            const uint32_t dg = (key[r] >> shift) & (WIDE_BINS - 1);
            const uint32_t lp = (uint32_t)L.wcnt[wave][dg] + pos[r];
            L.key[lp] = key[r];
            L.val[lp] = val[r];
        }
    }
    __syncthreads();
    // (the offsets are dead: their storage now holds the per-digit global bases)
#pragma unroll
    for (int j = 0; j < WIDE_DPT; ++j) L.glob[threadIdx.x * WIDE_DPT + j] = gfirst[j];
    __syncthreads();
    const uint32_t nvalid = L.nvalid;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const uint32_t lp = (uint32_t)j * SORT_THREADS + threadIdx.x;
        if (lp < nvalid) {
            const uint32_t k = L.key[lp];
            const uint32_t dst = L.glob[(k >> shift) & (WIDE_BINS - 1)] + lp;
            if (!NOKEYS) keys_out[dst] = k;
            vals_out[dst] = L.val[lp];
        }
    }
}

// Stable sort of n 32-bit keys (all 32 bits) with the identity payload: keys equal to 0xFFFFFFFF are dropped, the kept
// count goes to *kept_dev, and the sorted payload lands in `v0` (three passes: v0 -> v1 -> v0).  k0 holds the keys on
// entry; k0 / k1 are clobbered and the sorted keys are not produced (nobody reads them).  scratch: depth_sort_scratch_words(n).
int launch_depth_sort_wide(uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, size_t n, uint32_t* scratch,
                           hipStream_t s, uint32_t* kept_dev) {
    if (n == 0) return 0;
    const unsigned nb = (unsigned)sort_blocks(n);
    const size_t hn = (size_t)nb * WIDE_BINS;
    uint32_t* hist = scratch;
    uint32_t* scan_scratch = scratch + hn + (hn & 1);
    unsigned long long* desc = reinterpret_cast<unsigned long long*>(scan_scratch);
    const unsigned ndesc = (unsigned)scan_blocks(hn) + 1u;
    const uint32_t* n_dev = nullptr;
    uint32_t *ki = k0, *ko = k1, *vi = nullptr, *vo = v0;
    for (int p = 0; p < 3; ++p) {
        const int shift = WIDE_BITS * p;
        if (p == 0) wide_hist_kernel<true><<<dim3(nb), dim3(SORT_THREADS), 0, s>>>(ki, n, n_dev, shift, hist, nb, desc, ndesc);
        else wide_hist_kernel<false><<<dim3(nb), dim3(SORT_THREADS), 0, s>>>(ki, n, n_dev, shift, hist, nb, desc, ndesc);
        LAUNCH_OK("wide_hist_kernel");
        const int rc = launch_scan_chained_u32(hist, hist, hn, scan_scratch, false, s, nullptr, p == 0 ? kept_dev : nullptr);
        if (rc) return rc;
        if (p == 0) wide_scatter_kernel<true, false><<<dim3(nb), dim3(SORT_THREADS), 0, s>>>(ki, vi, ko, vo, n, n_dev, shift, hist, nb);
        else if (p == 1) wide_scatter_kernel<false, false><<<dim3(nb), dim3(SORT_THREADS), 0, s>>>(ki, vi, ko, vo, n, n_dev, shift, hist, nb);
        else wide_scatter_kernel<false, true><<<dim3(nb), dim3(SORT_THREADS), 0, s>>>(ki, vi, ko, vo, n, n_dev, shift, hist, nb);
        LAUNCH_OK("wide_scatter_kernel");
        n_dev = kept_dev;
        uint32_t* t = ki; ki = ko; ko = t;
        vi = vo; vo = (vo == v0) ? v1 : v0;
    }
    return 0;
}

// workgroups of a segment-aligned (RANGES) pass over n keys
static inline unsigned seg_blocks(size_t n) { return (unsigned)(sort_blocks(n) + 256); }

template <typename KeyT>
static int radix_sort_pairs_t(KeyT* k0, KeyT* k1, uint32_t* v0, uint32_t* v1, size_t n, const uint32_t* n_dev_in, int nbits,
                              uint32_t* scratch, KeyT** keys_out, uint32_t** vals_out, hipStream_t s, bool identity_payload,
                              uint32_t* drop_count_dev, uint2* ranges_out, uint32_t nranges, uint32_t* lpt_cnt,
                              uint32_t* lpt_list) {
    KeyT *ki = k0, *ko = k1;
    uint32_t *vi = v0, *vo = v1;
    if (identity_payload && nbits < 1) nbits = 1;        // the payload (0, 1, 2, ...) only exists after a pass
    const int passes = radix_passes(nbits);
    const bool fuse_ranges = ranges_out && passes == 2;
    const uint32_t* n_dev = n_dev_in;
    if (n > 0) {
        // scratch: [hist: 256 x workgroups (segment-aligned upper bound)][seg: 257 (+pad)][scan descriptors (64-bit)]
        const unsigned nb = (unsigned)sort_blocks(n), nb_seg = seg_blocks(n);
        const size_t hn_max = (size_t)nb_seg * 256;
        uint32_t* hist = scratch;
        uint32_t* seg = scratch + hn_max;
        uint32_t* scan_scratch = seg + 260;               // (8-byte aligned: hn_max and 260 are even)
        for (int p = 0; p < passes; ++p) {
            const int shift = 8 * p;
            const bool first = p == 0, last = p == passes - 1;
            const bool drop = first && drop_count_dev != nullptr;
            const bool ranges = last && fuse_ranges;
            const unsigned blocks = ranges ? nb_seg : nb;
            const size_t hn = (size_t)blocks * 256;
            const unsigned ndesc = (unsigned)scan_blocks(hn) + 1u;
            unsigned long long* desc = reinterpret_cast<unsigned long long*>(scan_scratch);
            if (ranges)
                radix_hist_seg_kernel<KeyT><<<dim3(blocks), dim3(SORT_THREADS), 0, s>>>(ki, seg, shift, hist, blocks, desc, ndesc);
            else if (drop)
                radix_hist_kernel<KeyT, true><<<dim3(blocks), dim3(SORT_THREADS), 0, s>>>(ki, n, n_dev, shift, hist, blocks, desc, ndesc);
            else
                radix_hist_kernel<KeyT, false><<<dim3(blocks), dim3(SORT_THREADS), 0, s>>>(ki, n, n_dev, shift, hist, blocks, desc, ndesc);
            LAUNCH_OK("radix_hist_kernel");
            // (a compacting pass: the scan's grand total is the kept count)
            int rc = launch_scan_chained_u32(hist, hist, hn, scan_scratch, false, s, nullptr, drop ? drop_count_dev : nullptr);
            if (rc) return rc;
            const uint32_t* vin = (identity_payload && first) ? nullptr : vi;
            // (the pass in front of a RANGES pass leaves the boundaries of ITS digit's segments in `seg`)
            uint32_t* seg_arg = (ranges || (fuse_ranges && p == passes - 2)) ? seg : nullptr;
            if (ranges)
                radix_scatter_kernel<KeyT, false, true, true><<<dim3(blocks), dim3(SORT_THREADS), 0, s>>>(
                    ki, vin, ko, vo, n, n_dev, shift, hist, blocks, seg_arg, ranges_out, nranges, lpt_cnt, lpt_list);
            else if (drop)
                radix_scatter_kernel<KeyT, true, false, false><<<dim3(blocks), dim3(SORT_THREADS), 0, s>>>(
                    ki, vin, ko, vo, n, n_dev, shift, hist, blocks, seg_arg, nullptr, 0u, nullptr, nullptr);
            else
                radix_scatter_kernel<KeyT, false, false, false><<<dim3(blocks), dim3(SORT_THREADS), 0, s>>>(
                    ki, vin, ko, vo, n, n_dev, shift, hist, blocks, seg_arg, nullptr, 0u, nullptr, nullptr);
            LAUNCH_OK("radix_scatter_kernel");
            if (drop) n_dev = drop_count_dev;           // the later passes sort the kept keys only
            KeyT* t = ki; ki = ko; ko = t;
            uint32_t* u = vi; vi = vo; vo = u;
        }
        if (ranges_out && !fuse_ranges) {
            tile_ranges_kernel<KeyT><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(n_dev, (uint32_t)n, ki, ranges_out, nranges);
            LAUNCH_OK("tile_ranges_kernel");
        }
    }
    *keys_out = fuse_ranges ? nullptr : ki;         // (a RANGES pass writes no keys)
    *vals_out = vi;
    return 0;
}

int launch_radix_sort_pairs(uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, size_t n, int nbits,
                            uint32_t* scratch, uint32_t** keys_out, uint32_t** vals_out, hipStream_t s,
                            bool identity_payload, uint32_t* drop_count_dev, const uint32_t* n_dev, uint2* ranges_out,
                            uint32_t nranges, uint32_t* lpt_cnt, uint32_t* lpt_list) {
    return radix_sort_pairs_t<uint32_t>(k0, k1, v0, v1, n, n_dev, nbits, scratch, keys_out, vals_out, s, identity_payload,
                                        drop_count_dev, ranges_out, nranges, lpt_cnt, lpt_list);
}

int launch_radix_sort_pairs_u16(uint16_t* k0, uint16_t* k1, uint32_t* v0, uint32_t* v1, size_t n, int nbits,
                                uint32_t* scratch, uint16_t** keys_out, uint32_t** vals_out, hipStream_t s,
                                bool identity_payload, const uint32_t* n_dev, uint2* ranges_out, uint32_t nranges,
                                uint32_t* lpt_cnt, uint32_t* lpt_list) {
    return radix_sort_pairs_t<uint16_t>(k0, k1, v0, v1, n, n_dev, nbits > 16 ? 16 : nbits, scratch, keys_out, vals_out, s,
                                        identity_payload, nullptr, ranges_out, nranges, lpt_cnt, lpt_list);
}
