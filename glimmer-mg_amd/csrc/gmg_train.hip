// gmg_train.hip -- the counting side of build-icm on the device.
//
// The reference trains an ICM level by level (src/ICM/icm.cc:1356-1455 Train_Model, 1061-1186 Complete_Tree): for
// every level it walks every window of every training string down the tree built so far (Get_Training_Node,
// icm.cc:1233-1256) and adds the window's model_len - 1 (context base, predicted base) pairs to the 4 x 4 tables of
// the node it lands on (Count_Char_Pairs_Restricted, icm.cc:1190-1229; Count_Char_Pairs, icm.cc:1841-1870, for the
// roots).  That is all the work that grows with the training set; picking the context position from the tables and the
// chi-squared interpolation are per node and stay host C++ (glimmer-mg_amd/host/icm_train.cc), as libm's log decides
// ties there.
//
// Here: one lane per window.  The node a window reached at level l - 1 is kept in HBM (4 bytes per window), so a level
// is ONE step of the descent instead of l, a coalesced 4-byte read + write per window, the window's 2-bit codes from
// the packed stream, and model_len - 1 integer increments.  Where the increments go decides the speed:
//   * levels whose tables fit 64 KB of LDS (the top three for the default 12 / 7 / 3 shape) are counted per workgroup
//     in LDS and flushed once;
//   * deeper levels: device-wide atomics from 8 XCDs are served behind the L2s, one memory-side transaction per
//     increment (27 G increments/s measured, 26 - 38 ms per level for 63 M windows).  So the windows of the level
//     are sorted by table first (hipcub radix sort of (table, value) pairs, <= 17 key bits; the value is the window's
//     codes when they fit 32 bits, else its position); a workgroup then takes 4,096 consecutive pairs, which touch
//     a handful of tables, counts them in LDS and adds each table's 176 counters to HBM once (1.65 ms per level
//     for 63 M windows: 0.40 step + 0.98 sort + 0.26 count);
//   * tiny training sets (below GMG_TRAIN_SORT_MIN bases, default 2^16) keep the direct atomics: five more launches
//     per level would cost more than they save.  (One genome's genes, 1.6 M windows: 0.77 ms per deep level with
//     direct atomics, 0.14 - 0.26 ms sorted.)
//
//   k_train_level<LDS, KEYS, NT>   the descent step + counts of one level (LDS / global atomics) or + the sort keys
//   k_train_count_sorted           counts from the sorted (table, value) pairs
//
// Integer work only: counts are exact, so the tables equal the reference's for any order of the atomics.

#include "gmg_device.h"

#include <hipcub/hipcub.hpp>

#include <new>
#include <stdlib.h>
#include <stdio.h>
#include <vector>

struct gmg_trainer {
    const gmg_reads *strings;   // borrowed: must outlive the trainer
    int W, D, P;
    int next_level;             // levels are taken in order
    int32_t *d_node;            // [total_bases] node reached by the window starting at that base, -1 = none / stopped
    int8_t *d_mip_prev;         // [P][4^(level-1)] mut_info_pos of the level above the one being counted
    int32_t *d_cnt;             // [P][4^level][max(W-1,1)][16] of the level being counted
    size_t cnt_cap;             // counters allocated at d_cnt
    // the sorted path (allocated on first use): (table, window) pairs before / after the sort + hipcub's scratch
    uint32_t *d_key, *d_key_sorted, *d_win, *d_win_sorted;
    void *d_sort_tmp;
    size_t sort_tmp_bytes;
};

namespace {

struct TrainArgs {
    const uint32_t *packed;
    const uint64_t *off;
    const uint32_t *tile_read;
    uint64_t total_bases, n_tiles;
    int W, P, npos;             // npos = max(W - 1, 1)
    int level, first, on_level; // first node of the level, nodes on it
    int prev_first, prev_on;    // the level above
    const int8_t *mip_prev;
    int32_t *node;
    int32_t *cnt;
    uint32_t cnt_len;           // P * on_level * npos * 16
    uint32_t *key, *win;        // KEYS: table of window g (P * on_level = none) and its codes (W <= 16) or g itself
};

template <bool LDS, bool KEYS, int NT>
__global__ __launch_bounds__(NT) void k_train_level(TrainArgs a)
{
    extern __shared__ int32_t hist[];
    if (LDS) {
        for (uint32_t i = threadIdx.x; i < a.cnt_len; i += NT) hist[i] = 0;
        __syncthreads();
    }
    int32_t *const dst = LDS ? hist : a.cnt;

    for (uint64_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        const uint32_t r_lo = a.tile_read[t], r_hi = a.tile_read[t + 1];
#pragma unroll
        for (int k = 0; k < GMG_TILE / NT; k++) {
            const uint64_t g = t * GMG_TILE + (uint64_t)k * NT + threadIdx.x;
            if (g >= a.total_bases) break;
            // string holding base g: the largest r in [r_lo, r_hi] with off[r] <= g (empty strings share offsets)
            uint32_t lo = r_lo, hi = r_hi;
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if (a.off[mid] <= g) lo = mid; else hi = mid - 1;
            }
            const uint64_t begin = a.off[lo], end = a.off[lo + 1];
            // frame of the window that starts at offset s of its string: (model_len + s) mod periodicity
            // (icm.cc:1203-1226; Train_Model's per-frame offsets, icm.cc:1373-1375, say the same for the roots)
            const int frame = (int)(((uint64_t)a.W + (g - begin)) % (uint64_t)a.P);
            int node = -1;
            if (g + (uint64_t)a.W <= end) {                                  // a complete window
                if (a.level == 0) node = 0;
                else {
                    const int prev = a.node[g];
                    if (prev >= 0) {
                        const int p = a.mip_prev[frame * a.prev_on + (prev - a.prev_first)];
                        if (p >= 0) node = 4 * prev + dev_code(a.packed, g + (uint64_t)p) + 1;   // icm.cc:1246-1252
                    }
                }
            }
            a.node[g] = node;
            if (KEYS) {                                                      // counted after the sort
                a.key[g] = node < 0 ? (uint32_t)(a.P * a.on_level) : (uint32_t)(frame * a.on_level + (node - a.first));
                // the pair's value: the window's codes themselves when they fit 32 bits (W <= 16) -- the count kernel then
                // has nothing to gather (63 M random 12-byte reads per level were its bound) -- else the window's position
                a.win[g] = a.W <= 16 ? (node < 0 ? 0u : (uint32_t)dev_window_bits(a.packed, (int64_t)g)) : (uint32_t)g;
                continue;
            }
            if (node < 0) continue;
            const uint64_t bits = dev_window_bits(a.packed, (int64_t)g);     // 32 bases from g on
            const uint32_t last = (uint32_t)(bits >> (2 * (a.W - 1))) & 3u;
            int32_t *ct = dst + ((size_t)(frame * a.on_level + (node - a.first)) * a.npos) * 16 + last;
            if (a.W == 1) { atomicAdd(ct, 1); continue; }                    // Count_Single_Chars, icm.cc:1874-1896
            uint64_t b = bits;
            for (int i = 0; i < a.W - 1; i++, b >>= 2) atomicAdd(ct + i * 16 + 4 * (int)(b & 3u), 1);   // icm.cc:1214-1219
        }
    }

    if (LDS) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < a.cnt_len; i += NT) {
            const int32_t v = hist[i];
            if (v) atomicAdd(a.cnt + i, v);
        }
    }
}

// Counts from (table, window) pairs sorted by table.  A workgroup takes SORT_CHUNK consecutive pairs; the tables
// k0 .. k0 + SORT_TABLES - 1 (k0 = the chunk's first) are counted in LDS, anything beyond goes to HBM directly
// (only where tables hold a handful of windows each).
#define SORT_CHUNK 4096
#define SORT_NT 1024
#define SORT_LDS_INTS (12 * 1024)            // 48 KB
__global__ __launch_bounds__(SORT_NT) void k_train_count_sorted(const uint32_t *__restrict__ key, const uint32_t *__restrict__ win,
                                                                uint64_t n, const uint32_t *__restrict__ packed, int W, int npos,
                                                                uint32_t n_tables, int32_t *__restrict__ cnt)
{
    __shared__ int32_t hist[SORT_LDS_INTS];
    const uint64_t base = (uint64_t)blockIdx.x * SORT_CHUNK;
    const uint32_t k0 = key[base];
    if (k0 >= n_tables) return;                                              // only windows that count nothing from here on
    const uint32_t tbl = (uint32_t)npos * 16;
    // tables held in LDS: as many as fit, but no more than the chunk spans (the pairs are sorted: its last pair names
    // the last table) -- with 10^3 windows per table that is a handful, and zeroing / flushing 48 KB per 4,096 windows
    // would cost as much as the counting
    const uint64_t last_e = (base + SORT_CHUNK < n ? base + SORT_CHUNK : n) - 1;
    const uint32_t k_last = key[last_e] < n_tables ? key[last_e] : n_tables - 1;
    const uint32_t fit = SORT_LDS_INTS / tbl, span = k_last - k0 + 1;
    const uint32_t in_lds = span < fit ? span : fit;
    const uint32_t used = in_lds * tbl;
    constexpr int PER = SORT_CHUNK / SORT_NT;
    // all of a lane's pairs and their windows' codes are fetched before the first increment
    uint32_t k[PER];
    uint64_t bits[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint64_t e = base + (uint64_t)j * SORT_NT + threadIdx.x;
        k[j] = e < n ? key[e] : n_tables;
        const uint32_t w = e < n ? win[e] : 0u;
        bits[j] = W <= 16 ? (uint64_t)w : dev_window_bits(packed, (int64_t)w);
    }
    for (uint32_t i = threadIdx.x; i < used; i += SORT_NT) hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; j++) {
        if (k[j] >= n_tables) continue;
        const uint32_t last = (uint32_t)(bits[j] >> (2 * (W - 1))) & 3u;
        const uint32_t rel = k[j] - k0;
        uint64_t b = bits[j];
        if (rel < in_lds) {                                                  // ds_add_u32
            int32_t *ct = hist + rel * tbl + last;
            if (W == 1) atomicAdd(ct, 1);
            else   // neighbouring lanes share the table: each starts at its own context position, so that one ds_add
                   // instruction spreads over the W - 1 sixteen-counter blocks of the table instead of one
                for (int s = 0, i = (int)(threadIdx.x % (unsigned)(W - 1)); s < W - 1; s++, i = (i + 1 == W - 1 ? 0 : i + 1))
                    atomicAdd(ct + i * 16 + 4 * (int)((b >> (2 * i)) & 3u), 1);
        } else {
            int32_t *ct = cnt + (size_t)k[j] * tbl + last;
            if (W == 1) atomicAdd(ct, 1);
            else for (int i = 0; i < W - 1; i++, b >>= 2) atomicAdd(ct + i * 16 + 4 * (int)(b & 3u), 1);
        }
    }
    __syncthreads();
    const size_t out0 = (size_t)k0 * tbl, out_end = (size_t)n_tables * tbl;
    for (uint32_t i = threadIdx.x; i < used && out0 + i < out_end; i += SORT_NT) {
        const int32_t v = hist[i];
        if (v) atomicAdd(cnt + out0 + i, v);
    }
}

size_t sort_min_bases(void)
{
    const long long v = gmg_opt(GMG_OPT_TRAIN_SORT_MIN);          // gmg_set_option("train_sort_min", ..): tests switch paths
    return v >= 0 ? (size_t)v : (size_t)1 << 16;
}

int level_first(int level)   // (4^level - 1) / 3
{
    int pw = 1;
    for (int i = 0; i < level; i++) pw *= 4;
    return (pw - 1) / 3;
}

}  // namespace

extern "C" int gmg_trainer_create(const gmg_reads *strings, int model_len, int model_depth, int periodicity,
                                  gmg_trainer **out)
{
    { int rc_enter = gmg_enter("gmg_trainer_create"); if (rc_enter) return rc_enter; }
    if (!strings || !out) return gmg_set_error(GMG_EINVAL, "gmg_trainer_create: NULL argument");
    if (model_len < 1 || model_len > GMG_MAX_MODEL_LEN || model_depth < 0 || model_depth > 12 || periodicity < 1)
        return gmg_set_error(GMG_EBADMODEL, "gmg_trainer_create: model_len %d (1..%d), model_depth %d (0..12), "
                             "periodicity %d (>= 1)", model_len, GMG_MAX_MODEL_LEN, model_depth, periodicity);
    {   // the counters of the deepest level are indexed with 32 bits
        const double counters = (double)periodicity * (double)(level_first(model_depth + 1) - level_first(model_depth)) *
                                (model_len > 1 ? model_len - 1 : 1) * 16.0;
        if (counters >= 2147483648.0)
            return gmg_set_error(GMG_EBADMODEL, "gmg_trainer_create: %.0f counters on level %d (model_len %d, periodicity %d); "
                                 "at most 2^31 - 1", counters, model_depth, model_len, periodicity);
    }
    gmg_trainer *t = new (std::nothrow) gmg_trainer();
    if (!t) return gmg_set_error(GMG_ENOMEM, "gmg_trainer_create: out of host memory");
    t->strings = strings;
    t->W = model_len;
    t->D = model_depth;
    t->P = periodicity;
    t->next_level = 0;
    t->d_node = nullptr;
    t->d_mip_prev = nullptr;
    t->d_cnt = nullptr;
    t->d_key = t->d_key_sorted = t->d_win = t->d_win_sorted = nullptr;
    t->d_sort_tmp = nullptr;
    t->sort_tmp_bytes = 0;
    const int npos = model_len > 1 ? model_len - 1 : 1;
    const size_t on_last = (size_t)(level_first(model_depth + 1) - level_first(model_depth));
    t->cnt_cap = (size_t)periodicity * on_last * npos * 16;
    const size_t prev_cap = (size_t)periodicity * (model_depth > 0 ? on_last / 4 : 1);
    // from the library's cache of device blocks: a training run per genome asks for the same sizes again and again
    hipError_t e = gmg_pool_alloc((void **)&t->d_node, (strings->total_bases + 1) * sizeof(int32_t));
    if (e == hipSuccess) e = gmg_pool_alloc((void **)&t->d_mip_prev, prev_cap);
    if (e == hipSuccess) e = gmg_pool_alloc((void **)&t->d_cnt, t->cnt_cap * sizeof(int32_t));
    if (e != hipSuccess) {
        gmg_trainer_free(t);
        return gmg_set_error(GMG_ENOMEM, "gmg_trainer_create: device allocation failed: %s", hipGetErrorString(e));
    }
    *out = t;
    return GMG_OK;
}

extern "C" int gmg_trainer_free(gmg_trainer *t)
{
    if (!t) return GMG_OK;
    (void)hipDeviceSynchronize();
    if (t->d_node) gmg_pool_release(t->d_node);
    if (t->d_mip_prev) gmg_pool_release(t->d_mip_prev);
    if (t->d_cnt) gmg_pool_release(t->d_cnt);
    gmg_pool_release(t->d_key);
    gmg_pool_release(t->d_key_sorted);
    gmg_pool_release(t->d_win);
    gmg_pool_release(t->d_win_sorted);
    gmg_pool_release(t->d_sort_tmp);
    delete t;
    return GMG_OK;
}

extern "C" int gmg_trainer_level_counts(gmg_trainer *t, int level, const int16_t *mip_prev, int32_t *counts)
{
    { int rc_enter = gmg_enter("gmg_trainer_level_counts"); if (rc_enter) return rc_enter; }
    if (!t || !counts) return gmg_set_error(GMG_EINVAL, "gmg_trainer_level_counts: NULL argument");
    if (level != t->next_level || level > t->D)
        return gmg_set_error(GMG_EINVAL, "gmg_trainer_level_counts: level %d asked, level %d is next (levels go in "
                             "order, 0 .. model_depth = %d)", level, t->next_level, t->D);
    if (level > 0 && !mip_prev) return gmg_set_error(GMG_EINVAL, "gmg_trainer_level_counts: mip_prev is NULL");
    const gmg_reads *r = t->strings;
    TrainArgs a;
    a.packed = r->d_packed;
    a.off = r->d_off;
    a.tile_read = r->d_tile_read;
    a.total_bases = r->total_bases;
    a.n_tiles = r->n_tiles;
    a.W = t->W;
    a.P = t->P;
    a.npos = t->W > 1 ? t->W - 1 : 1;
    a.level = level;
    a.first = level_first(level);
    a.on_level = level_first(level + 1) - a.first;
    a.prev_first = level > 0 ? level_first(level - 1) : 0;
    a.prev_on = level > 0 ? a.on_level / 4 : 1;
    a.mip_prev = t->d_mip_prev;
    a.node = t->d_node;
    a.cnt = t->d_cnt;
    const size_t cnt_len = (size_t)t->P * a.on_level * a.npos * 16;
    a.cnt_len = (uint32_t)cnt_len;
    a.key = a.win = nullptr;

    if (level > 0) {
        // the level above, narrowed to bytes; a context position outside the window would read another string
        const size_t n = (size_t)t->P * a.prev_on;
        std::vector<int8_t> m8(n);
        for (size_t i = 0; i < n; i++) {
            if (mip_prev[i] < -2 || mip_prev[i] > t->W - 2)
                return gmg_set_error(GMG_EBADMODEL, "gmg_trainer_level_counts: mut_info_pos %d outside [-2, %d]",
                                     (int)mip_prev[i], t->W - 2);
            m8[i] = (int8_t)mip_prev[i];
        }
        GMG_HIP(hipMemcpy(t->d_mip_prev, m8.data(), n, hipMemcpyHostToDevice));
    }
    GMG_HIP(hipMemsetAsync(t->d_cnt, 0, cnt_len * sizeof(int32_t), 0));
    if (a.n_tiles > 0) {
        const bool lds = cnt_len * sizeof(int32_t) <= 64 * 1024;
        const bool sorted = !lds && a.total_bases >= sort_min_bases() && a.total_bases < 0x7fffffffull;
        if (lds) {
            // 1,024 lanes per workgroup: one table copy serves 16 waves (a 34 KB table x 256 lanes left 8 waves per CU)
            const unsigned grid = (unsigned)(a.n_tiles < 512 ? a.n_tiles : 512);
            hipLaunchKernelGGL((k_train_level<true, false, 1024>), dim3(grid), dim3(1024), cnt_len * sizeof(int32_t), 0, a);
        } else if (!sorted) {
            const unsigned grid = (unsigned)(a.n_tiles < 8192 ? a.n_tiles : 8192);
            hipLaunchKernelGGL((k_train_level<false, false, 256>), dim3(grid), dim3(256), 0, 0, a);
        } else {
            const uint64_t n = a.total_bases;
            const uint32_t n_tables = (uint32_t)(t->P * a.on_level);
            int end_bit = 1;
            while (end_bit < 32 && (1ull << end_bit) <= n_tables) end_bit++;     // the keys go up to n_tables (= no table)
            if (!t->d_key) {
                hipError_t e = gmg_pool_alloc((void **)&t->d_key, n * sizeof(uint32_t));
                if (e == hipSuccess) e = gmg_pool_alloc((void **)&t->d_key_sorted, n * sizeof(uint32_t));
                if (e == hipSuccess) e = gmg_pool_alloc((void **)&t->d_win, n * sizeof(uint32_t));
                if (e == hipSuccess) e = gmg_pool_alloc((void **)&t->d_win_sorted, n * sizeof(uint32_t));
                if (e != hipSuccess)
                    return gmg_set_error(GMG_ENOMEM, "gmg_trainer_level_counts: device allocation failed: %s", hipGetErrorString(e));
            }
            size_t need = 0;
            GMG_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, t->d_key, t->d_key_sorted, t->d_win, t->d_win_sorted,
                                                       (int)n, 0, end_bit, (hipStream_t)0));
            if (need > t->sort_tmp_bytes) {
                gmg_pool_release(t->d_sort_tmp);
                t->d_sort_tmp = nullptr;
                t->sort_tmp_bytes = 0;
                if (gmg_pool_alloc(&t->d_sort_tmp, need) != hipSuccess)
                    return gmg_set_error(GMG_ENOMEM, "gmg_trainer_level_counts: device allocation failed (sort scratch)");
                t->sort_tmp_bytes = need;
            }
            a.key = t->d_key;
            a.win = t->d_win;
            const unsigned grid = (unsigned)(a.n_tiles < 8192 ? a.n_tiles : 8192);
            hipLaunchKernelGGL((k_train_level<false, true, 256>), dim3(grid), dim3(256), 0, 0, a);
            GMG_HIP(hipGetLastError());
            need = t->sort_tmp_bytes;
            GMG_HIP(hipcub::DeviceRadixSort::SortPairs(t->d_sort_tmp, need, t->d_key, t->d_key_sorted, t->d_win, t->d_win_sorted,
                                                       (int)n, 0, end_bit, (hipStream_t)0));
            const unsigned cgrid = (unsigned)((n + SORT_CHUNK - 1) / SORT_CHUNK);
            hipLaunchKernelGGL(k_train_count_sorted, dim3(cgrid), dim3(SORT_NT), 0, 0, t->d_key_sorted, t->d_win_sorted, n,
                               a.packed, a.W, a.npos, n_tables, t->d_cnt);
        }
        GMG_HIP(hipGetLastError());
    }
    GMG_HIP(hipMemcpy(counts, t->d_cnt, cnt_len * sizeof(int32_t), hipMemcpyDeviceToHost));
    t->next_level = level + 1;
    return GMG_OK;
}
