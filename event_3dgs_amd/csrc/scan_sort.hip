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
#include <stdlib.h>

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

// ------------------------------------------------------------------------------------ single-launch scan
// The same scan in ONE launch (chained scan with decoupled look-back): for the scans that sit between two kernels of
// the list-building chain, where three launches of ~5 us each cost more than the scan itself.  Protocol as in
// radix_onesweep_kernel (cdna_hip_programming.md G16, form R2): one 32-bit descriptor {2-bit status, 30-bit sum} per
// workgroup, relaxed agent-scope atomics, logical workgroup ids from an atomic ticket.  Wave 0 of a workgroup looks
// back 64 predecessors at a time.  `desc` = nb + 1 words (descriptors, then the ticket) that MUST be zero on entry:
// the kernel in front of the scan in the stream zeroes them (radix_hist_kernel / preprocess_kernel), which costs
// no extra command.  Sums must stay below 2^30.
constexpr uint32_t SC_AGG = 1u << 30, SC_PREFIX = 2u << 30, SC_MASK = (1u << 30) - 1u;
template <bool INCLUSIVE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_chained_kernel(const uint32_t* in, uint32_t* out, size_t n,
                                                                    uint32_t* desc, unsigned nb,
                                                                    volatile int* total_host) {
    __shared__ uint32_t wtot[SCAN_THREADS / WAVE];
    __shared__ uint32_t s_bid, s_excl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_bid = atomicAdd(desc + nb, 1u);
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
            __hip_atomic_store(desc + bid, (bid == 0 ? SC_PREFIX : SC_AGG) | total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        int p = (int)bid - 1;                       // lane l looks at workgroup p - l
        while (p >= 0) {
            const int idx = p - lane;
            const uint32_t d = idx >= 0 ? __hip_atomic_load(desc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : SC_PREFIX;             // virtual workgroup -1: prefix 0
            const uint32_t st = d >> 30;
            const unsigned long long ready = __ballot(st != 0u), pre = __ballot(st == 2u);
            const int lead = (~ready) ? __builtin_ctzll(~ready) : 64;       // published entries in a row from lane 0
            const int firstpre = pre ? __builtin_ctzll(pre) : 64;
            const int use = firstpre < lead ? firstpre + 1 : lead;          // consumed this round
            uint32_t c = lane < use ? (d & SC_MASK) : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
            excl += c;
            if (firstpre < lead) break;                                     // reached an inclusive prefix
            p -= use;
            if (use == 0) __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) {
            if (bid != 0)
                __hip_atomic_store(desc + bid, SC_PREFIX | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    // the grand total straight into host-visible (pinned, mapped) memory, by the thread that owns the last element:
    // the host polls that word (E3_FLAG_COUNT_MAPPED) -- no separate publishing kernel, no copy command
    if (total_host && base < n && n <= base + SCAN_ITEMS) {
        *total_host = (int)run;
        __threadfence_system();
    }
}

void launch_scan_chained_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* desc_zeroed, bool inclusive,
                             hipStream_t s, int* total_host) {
    if (n == 0) return;
    const unsigned nb = (unsigned)scan_blocks(n);
    if (inclusive)
        scan_chained_kernel<true><<<dim3(nb), dim3(SCAN_THREADS), 0, s>>>(in, out, n, desc_zeroed, nb, total_host);
    else
        scan_chained_kernel<false><<<dim3(nb), dim3(SCAN_THREADS), 0, s>>>(in, out, n, desc_zeroed, nb, total_host);
}

// ------------------------------------------------------------------------------------ radix sort
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(const uint32_t* __restrict__ keys, size_t n,
                                                                  int shift, uint32_t* __restrict__ hist,
                                                                  unsigned nblocks, uint32_t* __restrict__ scan_desc,
                                                                  unsigned ndesc) {
    __shared__ uint32_t h[256];
    // (descriptors + ticket of the chained scan that follows in the stream: zeroed here instead of by a memset command)
    if (blockIdx.x == 0) for (unsigned t = threadIdx.x; t < ndesc; t += SORT_THREADS) scan_desc[t] = 0u;
    h[threadIdx.x] = 0;
    __syncthreads();
    size_t base = (size_t)blockIdx.x * SORT_TILE;
    // (all loads first, unconditional with a clamped index: a load inside `if (idx < n)` next to its use waits for its own
    // round trip in every unrolled iteration)
    uint32_t k[SORT_ITEMS];
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        size_t idx = base + (size_t)i * SORT_THREADS + threadIdx.x;
        k[i] = keys[idx < n ? idx : n - 1];
    }
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        size_t idx = base + (size_t)i * SORT_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&h[(k[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Final phase shared by both scatter kernels.  On entry wcnt[w][d] holds the per-wave digit counts, pos[r] the
// rank of each key inside its (wave, digit) group, and thread d knows `gfirst`, the global position of this
// workgroup's first key with digit d.  Keys are first placed at their workgroup-local sorted position in LDS
// and then streamed out by consecutive threads, so the lanes of a store instruction write consecutive
// addresses inside each digit's segment (avg 16 keys = 64 B) instead of 4-key fragments per (wave, digit).
template <int ITEMS>
struct StageLds {
    uint32_t key[SORT_THREADS * ITEMS];
    uint32_t val[SORT_THREADS * ITEMS];
    uint32_t glob[256];     // global position minus local position, per digit
    uint32_t wtot[SORT_THREADS / WAVE];
};
template <int ITEMS>
__device__ __forceinline__ void staged_scatter(const uint32_t (&key)[ITEMS], const uint32_t (&val)[ITEMS],
                                               const uint32_t (&pos)[ITEMS],
                                               uint32_t (*wcnt)[256], uint32_t gfirst, size_t block_first, size_t wbase,
                                               size_t n, int shift, uint32_t* __restrict__ keys_out,
                                               uint32_t* __restrict__ vals_out, StageLds<ITEMS>& L) {
    constexpr int TILE = SORT_THREADS * ITEMS;
    constexpr int NW = SORT_THREADS / WAVE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d = threadIdx.x;
    uint32_t c[NW], total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { c[w] = wcnt[w][d]; total += c[w]; }
    // workgroup-local exclusive scan of the digit totals
    uint32_t inc = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) L.wtot[wave] = inc;
    __syncthreads();
    uint32_t lbase = inc - total;
    for (int w = 0; w < wave; ++w) lbase += L.wtot[w];
    L.glob[d] = gfirst - lbase;
    uint32_t l = lbase;
#pragma unroll
    for (int w = 0; w < NW; ++w) { wcnt[w][d] = l; l += c[w]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        size_t idx = wbase + (size_t)r * WAVE + lane;
        if (idx < n) {
            uint32_t dg = (key[r] >> shift) & 255u;
            uint32_t lp = wcnt[wave][dg] + pos[r];
            L.key[lp] = key[r];
            L.val[lp] = val[r];
        }
    }
    __syncthreads();
    const uint32_t nvalid = (uint32_t)((n - block_first) < (size_t)TILE ? (n - block_first) : (size_t)TILE);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        uint32_t lp = (uint32_t)j * SORT_THREADS + threadIdx.x;
        if (lp < nvalid) {
            uint32_t k = L.key[lp];
            uint32_t dst = L.glob[(k >> shift) & 255u] + lp;
            keys_out[dst] = k;
            vals_out[dst] = L.val[lp];
        }
    }
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
    __shared__ StageLds<SORT_ITEMS> stage;
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
        val[r] = ok ? (vals_in ? vals_in[idx] : (uint32_t)idx) : 0u;   // no payload array: the payload is the index
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
    // digit = threadIdx.x: offs holds the global position of this workgroup's first key of each digit
    staged_scatter(key, val, pos, wcnt, offs[(size_t)threadIdx.x * nblocks + blockIdx.x],
                   (size_t)blockIdx.x * SORT_TILE, wbase, n, shift, keys_out, vals_out, stage);
}

// ------------------------------------------------------------------------------------ onesweep variant
// One launch per digit instead of five: the digit histograms of ALL passes come from one upfront kernel,
// and each pass fuses histogram + cross-workgroup prefix + scatter using a chained scan with decoupled
// look-back.  Inter-workgroup protocol (cdna_hip_programming.md G16, form R2): each (block, digit) descriptor
// is ONE 32-bit word {2-bit status, 30-bit count} written and read with relaxed agent-scope atomics
// (sc1: write-through / L1-bypass), so the data is its own flag and no fence is needed.  Logical block ids
// are handed out by an atomic ticket, so a block only ever waits for blocks that have already started
// (no dependence on dispatch order or placement).  All descriptor words are zeroed by a memset before the pass.
constexpr uint32_t OS_AGG = 1u << 30, OS_PREFIX = 2u << 30, OS_MASK = (1u << 30) - 1u;

// Keys per thread of the onesweep passes: the look-back chain is as long as the number of workgroups, so the depth
// sort (one launch per digit over a few million keys) runs with bigger tiles than the three-kernel passes (measured:
// 16 / 24 / 32 keys per thread -> 0.174 / 0.157 / 0.165 ms for the 3 M-key depth sort).
constexpr int OS_ITEMS = 24;
constexpr int OS_TILE = SORT_THREADS * OS_ITEMS;
static inline size_t onesweep_blocks(size_t n) { return (n + OS_TILE - 1) / OS_TILE; }

__global__ __launch_bounds__(SORT_THREADS) void radix_global_hist_kernel(const uint32_t* __restrict__ keys, size_t n,
                                                                         int passes, uint32_t* __restrict__ ghist) {
    __shared__ uint32_t h[4][256];
    for (int p = 0; p < passes; ++p) h[p][threadIdx.x] = 0;
    __syncthreads();
    size_t base = (size_t)blockIdx.x * OS_TILE;
    uint32_t ones = 0;
    uint32_t kk[OS_ITEMS];
#pragma unroll
    for (int i = 0; i < OS_ITEMS; ++i) {                       // loads first (see radix_hist_kernel)
        size_t idx = base + (size_t)i * SORT_THREADS + threadIdx.x;
        kk[i] = keys[idx < n ? idx : n - 1];
    }
#pragma unroll
    for (int i = 0; i < OS_ITEMS; ++i) {
        size_t idx = base + (size_t)i * SORT_THREADS + threadIdx.x;
        if (idx < n) {
            uint32_t k = kk[i];
            // (the all-ones key -- culled splats, a third of the depth keys -- would serialise 64 lanes on one LDS
            // counter in every pass: counted in a register instead)
            if (k == 0xFFFFFFFFu) ++ones;
            else for (int p = 0; p < passes; ++p) atomicAdd(&h[p][(k >> (8 * p)) & 255u], 1u);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ones += __shfl_xor(ones, o, 64);
    if ((threadIdx.x & 63) == 0 && ones)
        for (int p = 0; p < passes; ++p) atomicAdd(&h[p][255], ones);
    __syncthreads();
    for (int p = 0; p < passes; ++p) {
        uint32_t c = h[p][threadIdx.x];
        if (c) atomicAdd(&ghist[p * 256 + threadIdx.x], c);
    }
}

__global__ __launch_bounds__(SORT_THREADS) void radix_onesweep_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, size_t n, int shift, const uint32_t* __restrict__ ghist /*256, this pass*/,
    uint32_t* desc /* nblocks x 256, zeroed */, uint32_t* ticket /* zeroed */) {
    constexpr int NW = SORT_THREADS / WAVE;
    constexpr int WAVE_KEYS = OS_TILE / NW;
    __shared__ uint32_t wcnt[NW][256];
    __shared__ uint32_t wtot[NW];
    __shared__ uint32_t s_bid;
    __shared__ StageLds<OS_ITEMS> stage;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u);
#pragma unroll
    for (int w = 0; w < NW; ++w) wcnt[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t bid = s_bid;
    const size_t wbase = (size_t)bid * OS_TILE + (size_t)wave * WAVE_KEYS;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t key[OS_ITEMS], val[OS_ITEMS], pos[OS_ITEMS];
#pragma unroll
    for (int r = 0; r < OS_ITEMS; ++r) {
        size_t idx = wbase + (size_t)r * WAVE + lane;
        bool ok = idx < n;
        key[r] = ok ? keys_in[idx] : 0xFFFFFFFFu;
        val[r] = ok ? (vals_in ? vals_in[idx] : (uint32_t)idx) : 0u;
    }
#pragma unroll
    for (int r = 0; r < OS_ITEMS; ++r) {
        size_t idx = wbase + (size_t)r * WAVE + lane;
        bool ok = idx < n;
        uint32_t d = (key[r] >> shift) & 255u;
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
    uint32_t gfirst;
    {
        // digit = threadIdx.x
        const int d = threadIdx.x;
        uint32_t c[NW], total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { c[w] = wcnt[w][d]; total += c[w]; }
        uint32_t* my = desc + (size_t)bid * 256 + d;
        __hip_atomic_store(my, (bid == 0 ? OS_PREFIX : OS_AGG) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // exclusive prefix of this digit over the preceding logical blocks (decoupled look-back)
        // windowed: LB_WIN predecessors are fetched with independent loads, then consumed in order
        uint32_t excl = 0;
        constexpr int LB_WIN = 8;
        int p = (int)bid - 1;
        bool done_lb = p < 0;
        while (!done_lb) {
            uint32_t v[LB_WIN];
#pragma unroll
            for (int k = 0; k < LB_WIN; ++k)
                v[k] = (p - k >= 0) ? __hip_atomic_load(desc + (size_t)(p - k) * 256 + d, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT)
                                    : OS_PREFIX;            // virtual block -1: prefix 0
            int used = 0;
#pragma unroll
            for (int k = 0; k < LB_WIN; ++k) {
                if (done_lb || used != k) continue;
                const uint32_t st = v[k] >> 30;
                if (st == 0) continue;                      // not published yet: re-fetch from here
                excl += v[k] & OS_MASK;
                used = k + 1;
                if (st == 2) done_lb = true;
            }
            p -= used;
            if (!done_lb && used == 0) __builtin_amdgcn_s_sleep(1);
        }
        if (bid != 0)
            __hip_atomic_store(my, OS_PREFIX | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // exclusive scan of the global digit histogram over the 256 digits
        uint32_t gh = ghist[d], inc = gh;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t dbase = inc - gh;
        for (int w = 0; w < wave; ++w) dbase += wtot[w];
        gfirst = dbase + excl;
    }
    __syncthreads();
    staged_scatter(key, val, pos, wcnt, gfirst, (size_t)bid * OS_TILE, wbase, n, shift, keys_out, vals_out, stage);
}

static bool use_onesweep() {
    static int v = -1;
    // off by default since round 2: with the single-launch scan and the batched-load histogram a three-kernel pass of the
    // depth sort (hist 4 us + scan 6 us + scatter 12 us on 3 M pairs) beats the look-back chain of a onesweep pass (28 us +
    // its share of the global histogram) at every size measured; E3DGS_ONESWEEP=1 turns it back on
    if (v < 0) { const char* e = getenv("E3DGS_ONESWEEP"); v = (e && e[0] == '1') ? 1 : 0; }
    return v != 0;
}
static bool onesweep_small_keys() {     // tile-id sort (few bits, millions of pairs): classic passes by default
    static int v = -1;
    if (v < 0) { const char* e = getenv("E3DGS_ONESWEEP_TILE"); v = (e && e[0] == '1') ? 1 : 0; }
    return v != 0;
}
static size_t onesweep_max_blocks() {
    static long v = -1;
    if (v < 0) { const char* e = getenv("E3DGS_ONESWEEP_MAX_BLOCKS"); v = e ? atol(e) : 1024; }
    return (size_t)v;
}

static bool onesweep_chosen(size_t n, int nbits) {
    const int passes = radix_passes(nbits);
    return n > 0 && use_onesweep() && passes <= 4 &&
           (nbits == 32 ? onesweep_blocks(n) <= onesweep_max_blocks() : onesweep_small_keys());
}
size_t radix_sort_zero_words(size_t n, int nbits) {
    return onesweep_chosen(n, nbits) ? 1024 + 64 + (size_t)radix_passes(nbits) * onesweep_blocks(n) * 256 : 0;
}

void launch_radix_sort_pairs(uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, size_t n, int nbits,
                             uint32_t* scratch, uint32_t** keys_out, uint32_t** vals_out, hipStream_t s,
                             bool identity_payload, bool scratch_zeroed) {
    uint32_t *ki = k0, *ko = k1, *vi = v0, *vo = v1;
    if (identity_payload && nbits < 1) nbits = 1;        // the payload (0, 1, 2, ...) only exists after a pass
    const int passes = radix_passes(nbits);
    // onesweep wins while every workgroup is co-resident and the chain is short (depth sort of P Gaussians);
    // for the multi-million instance sort the plain three-kernel pass is faster on this chip
    if (onesweep_chosen(n, nbits)) {
        unsigned nb = (unsigned)onesweep_blocks(n);
        // scratch: [ghist 4*256][ticket 64 per pass ...][desc passes * nb * 256]
        uint32_t* ghist = scratch;
        uint32_t* tickets = scratch + 1024;
        uint32_t* desc = scratch + 1024 + 64;
        if (!scratch_zeroed) (void)hipMemsetAsync(scratch, 0, radix_sort_zero_words(n, nbits) * sizeof(uint32_t), s);
        radix_global_hist_kernel<<<dim3(nb), dim3(SORT_THREADS), 0, s>>>(ki, n, passes, ghist);
        for (int p = 0; p < passes; ++p) {
            radix_onesweep_kernel<<<dim3(nb), dim3(SORT_THREADS), 0, s>>>(ki, (identity_payload && p == 0) ? nullptr : vi,
                                                                         ko, vo, n, 8 * p, ghist + 256 * p,
                                                                         desc + (size_t)p * nb * 256, tickets + p);
            uint32_t* t = ki; ki = ko; ko = t;
            t = vi; vi = vo; vo = t;
        }
    } else if (n > 0) {
        unsigned nb = (unsigned)sort_blocks(n);
        size_t hn = (size_t)nb * 256;
        uint32_t* hist = scratch;
        uint32_t* scan_scratch = scratch + hn;
        for (int shift = 0; shift < nbits; shift += 8) {
            radix_hist_kernel<<<dim3(nb), dim3(SORT_THREADS), 0, s>>>(ki, n, shift, hist, nb, scan_scratch,
                                                                      (unsigned)scan_blocks(hn) + 1u);
            launch_scan_chained_u32(hist, hist, hn, scan_scratch, false, s);
            radix_scatter_kernel<<<dim3(nb), dim3(SORT_THREADS), 0, s>>>(
                ki, (identity_payload && shift == 0) ? nullptr : vi, ko, vo, n, shift, hist, nb);
            uint32_t* t = ki; ki = ko; ko = t;
            t = vi; vi = vo; vo = t;
        }
    }
    *keys_out = ki;
    *vals_out = vi;
}
