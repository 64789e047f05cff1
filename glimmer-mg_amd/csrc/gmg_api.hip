// gmg_api.hip -- extern "C" entry points declared in include/gmg.h.
// Host-side table flattening + handle management; all compute is in gmg_kernels.hip.
// No CPU fallback: without a gfx950 device every scoring entry point fails loudly.

#include "gmg_internal.h"
#include "gmg_scan.h"

#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";
static int g_device = -1;

int gmg_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *gmg_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------
// options (tuning / test switches): one table, set through the API or once from the environment at gmg_init
// ---------------------------------------------------------------------------
long long g_gmg_opt[GMG_OPT_COUNT] = {
    /* seg_plain */ 0, /* mg_tile */ 0, /* mg_one_stream */ 0, /* mg_err_flat */ 0, /* mg_err_calls */ 0, /* mg_err_calls_grow */ 0,
    /* orfs_exact_path */ 0, /* train_sort_min */ -1, /* mg_max_entries */ 0x7ffffffell, /* mg_timing */ 0, /* ingest_timing */ 0,
    /* train_timing */ 0, /* diag */ 0, /* strings_fused */ 1, /* mg_gene32 */ 1, /* mg_fused */ 1, /* mg_err_skip */ 1, /* mg_orfs_events */ 2,
    /* mg_err_tile */ -1, /* mg_err_tile_q */ 0, /* mg_err_qonly */ 1, /* mg_err_wave */ 1, /* mg_err_wave_q */ 0, /* orfs_walk8 */ 4, /* ingest_scans */ 0, /* ingest_piece_min */ 64ll << 20, /* mg_orfs_bits */ 0, /* orfs_q_poison */ 0};
static const char *const g_opt_name[GMG_OPT_COUNT] = {
    "seg_plain", "mg_tile", "mg_one_stream", "mg_err_flat", "mg_err_calls", "mg_err_calls_grow", "orfs_exact_path",
    "train_sort_min", "mg_max_entries", "mg_timing", "ingest_timing", "train_timing", "diag", "strings_fused",
    "mg_gene32", "mg_fused", "mg_err_skip", "mg_orfs_events", "mg_err_tile", "mg_err_tile_q", "mg_err_qonly", "mg_err_wave", "mg_err_wave_q", "orfs_walk8", "ingest_scans", "ingest_piece_min", "mg_orfs_bits", "orfs_q_poison"};

static int opt_index(const char *key)
{
    for (int i = 0; key && i < GMG_OPT_COUNT; i++)
        if (strcmp(key, g_opt_name[i]) == 0) return i;
    return -1;
}

extern "C" int gmg_set_option(const char *key, long long value)
{
    const int i = opt_index(key);
    if (i < 0) return gmg_set_error(GMG_EINVAL, "gmg_set_option: unknown option '%s'", key ? key : "(null)");
#ifndef GMG_ABLATIONS
    if (i == GMG_OPT_DIAG && value != 0)
        return gmg_set_error(GMG_EINVAL, "gmg_set_option: the ablation kernels are not in this build (-DGMG_ABLATIONS)");
#endif
    if (i == GMG_OPT_MG_MAX_ENTRIES && (value < 1 || value > 0x7ffffffell))     // n + 1 entries must fit the int of the hipcub calls
        return gmg_set_error(GMG_EINVAL, "gmg_set_option: mg_max_entries must be in [1, 2^31 - 2]");
    g_gmg_opt[i] = value;
    return GMG_OK;
}

extern "C" int gmg_get_option(const char *key, long long *value)
{
    const int i = opt_index(key);
    if (i < 0 || !value) return gmg_set_error(GMG_EINVAL, "gmg_get_option: unknown option '%s'", key ? key : "(null)");
    *value = g_gmg_opt[i];
    return GMG_OK;
}

// GMG_<NAME>=value in the environment, read ONCE (at gmg_init); a variable without a number counts as 1
static void options_from_env(void)
{
    for (int i = 0; i < GMG_OPT_COUNT; i++) {
        char name[64] = "GMG_";
        size_t k = 4;
        for (const char *c = g_opt_name[i]; *c && k + 1 < sizeof name; c++) name[k++] = (char)toupper((unsigned char)*c);
        name[k] = 0;
        const char *v = getenv(name);
        if (!v) continue;
        char *end = nullptr;
        long long x = strtoll(v, &end, 10);
        if (end == v) x = 1;
        (void)gmg_set_option(g_opt_name[i], x);
    }
}
extern "C" const char *gmg_version(void) { return "glimmer-mg_amd 0.1 (gfx950)"; }

extern "C" int gmg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int gmg_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return gmg_set_error(GMG_ENODEV, "gmg_init: no HIP device (%s); there is no CPU fallback",
                             e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= n)
        return gmg_set_error(GMG_EINVAL, "gmg_init: device %d out of range (have %d)", device, n);
    hipDeviceProp_t prop;
    GMG_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return gmg_set_error(GMG_ENODEV, "gmg_init: device %d is %s; this library is built for gfx950 only",
                             device, prop.gcnArchName);
    GMG_HIP(hipSetDevice(device));
    if (g_device < 0) options_from_env();
    g_device = device;
    return GMG_OK;
}

// HIP's current device is per host thread, g_device is per process: every entry point that allocates, copies or
// launches goes through here, so that a second host thread (a pipeline's fetcher, a worker of the caller) works on
// the device gmg_init chose and not on device 0.
int gmg_enter(const char *who)
{
    if (g_device < 0) return gmg_set_error(GMG_ENODEV, "%s: gmg_init() has not succeeded in this process", who);
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != g_device) GMG_HIP(hipSetDevice(g_device));   // (a thread-local read otherwise)
    return GMG_OK;
}

static int require_init(const char *who) { return gmg_enter(who); }

extern "C" int gmg_synchronize(void *stream)
{
    int rc = gmg_enter("gmg_synchronize");
    if (rc) return rc;
    GMG_HIP(hipStreamSynchronize((hipStream_t)stream));
    return GMG_OK;
}

// ---------------------------------------------------------------------------
// packing helpers (host only)
// ---------------------------------------------------------------------------

// tolower(Filter(ch)) then Subscript: src/Common/gene.cc:1139-1175, src/ICM/icm.cc:2008-2027.
// a c g t keep their meaning; r,d -> g; w,k -> t; everything else -> c.
extern "C" int gmg_base_code(int ch)
{
    switch (tolower(ch & 0xff)) {
    case 'a': return 0;
    case 'c': return 1;
    case 'g': case 'r': case 'd': return 2;
    case 't': case 'w': case 'k': return 3;
    default: return 1;
    }
}

extern "C" uint64_t gmg_packed_words(uint64_t total_bases) { return (total_bases + 15) / 16 + 1; }

// gmg_base_code of every byte value, built on first use
static const uint8_t *base_code_table(void)
{
    static const struct Table {
        uint8_t code[256];
        Table() { for (int c = 0; c < 256; c++) code[c] = (uint8_t)gmg_base_code(c); }
    } table;
    return table.code;
}

extern "C" int gmg_pack_bases(const char *ascii, uint64_t n, uint64_t first_base, uint32_t *packed)
{
    if ((!ascii && n) || !packed) return gmg_set_error(GMG_EINVAL, "gmg_pack_bases: NULL argument");
    const uint8_t *code = base_code_table();
    const unsigned char *a = (const unsigned char *)ascii;
    uint64_t i = 0, g = first_base;
    for (; i < n && (g & 15); i++, g++) packed[g >> 4] |= (uint32_t)code[a[i]] << (2 * (g & 15));   // up to a word boundary
    for (; i + 16 <= n; i += 16, g += 16) {                                                          // whole words
        uint32_t w = 0;
        for (int k = 0; k < 16; k++) w |= (uint32_t)code[a[i + k]] << (2 * k);
        packed[g >> 4] |= w;
    }
    for (; i < n; i++, g++) packed[g >> 4] |= (uint32_t)code[a[i]] << (2 * (g & 15));
    return GMG_OK;
}

// ---------------------------------------------------------------------------
// models
// ---------------------------------------------------------------------------

static inline int parent_of(int x) { return (x - 1) / 4; }   // src/ICM/icm.hh:84, C truncation

// Row used when a descent (full or partial window) ends at original node n: a cut node
// (mip -2) hands over to its parent (src/ICM/icm.cc:590-595, 834-835).  mip < -2 is rejected at
// upload, and a reachable cut node always has a parent with mip >= 0, so the second look the
// full-window code takes (icm.cc:577-583 then 590) can never move further up.
static int final_row(const int16_t *mip, int n) { return mip[n] == -2 ? parent_of(n) : n; }

// Expand one sub-model into the completed tree (see gmg_internal.h): breadth-first; per node either
// the original node it mirrors (>= 0) or ~row once the original descent has stopped above it.
static void complete_tree(const int16_t *mip, const float *prob, int D, uint8_t *cshift, float *crow)
{
    std::vector<int> cur(1, 0), nxt;
    size_t lvl_base = 0, lvl_size = 1;
    for (int l = 0; l <= D; l++) {
        if (l < D) nxt.assign(lvl_size * 4, 0);
        for (size_t i = 0; i < lvl_size; i++) {
            int st = cur[i];
            int row = (st >= 0) ? final_row(mip, st) : ~st;
            memcpy(crow + 4 * (lvl_base + i), prob + 4 * (size_t)row, 4 * sizeof(float));
            if (l == D) continue;
            uint8_t sh = 0;
            if (st >= 0 && mip[st] >= 0) {
                sh = (uint8_t)(2 * mip[st]);
                for (int b = 0; b < 4; b++) nxt[4 * i + b] = 4 * st + 1 + b;
            } else {
                for (int b = 0; b < 4; b++) nxt[4 * i + b] = ~row;
            }
            cshift[lvl_base + i] = sh;
        }
        lvl_base += lvl_size;
        lvl_size *= 4;
        cur.swap(nxt);
    }
}

// Full-window value for window index idx (code of w[k] at bits 2k), by the plain descent.
static float dense_entry(const int16_t *mip, const float *prob, int W, int D, uint32_t idx)
{
    int node = 0;
    for (int i = 0; i < D; i++) {
        int pos = mip[node];
        if (pos == -1) break;
        if (pos < -1) { node = parent_of(node); break; }
        node = 4 * node + (int)((idx >> (2 * pos)) & 3) + 1;
    }
    if (mip[node] < -1) node = parent_of(node);
    return prob[4 * (size_t)node + ((idx >> (2 * (W - 1))) & 3)];
}

// Partial-window value for buffer position j < W-1 (src/ICM/icm.cc:807-842); idx holds B[i] at bits 2i.
static float dense_part_entry(const int16_t *mip, const float *prob, int W, int D, int j, uint32_t idx)
{
    int node = 0, start = j - (W - 1);
    for (int i = 0; i < D; i++) {
        int q = start + mip[node];
        if (q < 0) break;
        node = 4 * node + (int)((idx >> (2 * q)) & 3) + 1;
    }
    if (mip[node] == -2) node = parent_of(node);
    return prob[4 * (size_t)node + ((idx >> (2 * j)) & 3)];
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" int gmg_model_upload(const int16_t *mip, const float *prob4, int W, int D, int P, int N,
                                gmg_model **out)
{
    int rc = require_init("gmg_model_upload");
    if (rc) return rc;
    if (!mip || !prob4 || !out) return gmg_set_error(GMG_EINVAL, "gmg_model_upload: NULL argument");
    if (W < 1 || W > GMG_MAX_MODEL_LEN || D < 0 || D > 12 || P < 1 || N < 1)
        return gmg_set_error(GMG_EBADMODEL, "gmg_model_upload: unsupported shape len=%d depth=%d period=%d nodes=%d",
                             W, D, P, N);
    long need = 0, pw = 1;
    for (int l = 0; l <= D; l++) { need += pw; pw *= 4; }
    if (N < need)
        return gmg_set_error(GMG_EBADMODEL, "gmg_model_upload: num_nodes=%d < %ld needed for depth %d", N, need, D);
    const size_t PN = (size_t)P * N;
    for (size_t i = 0; i < PN; i++)
        if (mip[i] < -2 || mip[i] > W - 1)
            return gmg_set_error(GMG_EBADMODEL, "gmg_model_upload: mut_info_pos %d at slot %zu outside [-2,%d]",
                                 (int)mip[i], i, W - 1);

    const bool fast = (W <= GMG_FAST_MAX_LEN && D <= GMG_FAST_MAX_DEPTH);
    const bool dense = (W <= GMG_DENSE_MAX_LEN);
    const size_t n_internal = (size_t)((pw / 4 - 1) / 3);   // (4^D - 1) / 3
    const size_t ctot = (size_t)need;                        // (4^(D+1) - 1) / 3
    const size_t cstride = align_up(n_internal ? n_internal : 1, 16);
    const size_t n_dense = dense ? ((size_t)1 << (2 * W)) : 0;
    const size_t n_part = dense ? (n_dense - 4) / 3 : 0;     // sum_{j<W-1} 4^(j+1)

    size_t o_mip = 0;
    size_t o_prob = align_up(o_mip + PN, 256);
    size_t o_cshift = align_up(o_prob + PN * 16, 256);
    size_t o_crow = align_up(o_cshift + (fast ? P * cstride : 0), 256);
    size_t o_chalf = align_up(o_crow + (fast ? (size_t)P * ctot * 16 + 16 : 0), 256);   // + one all-zero row
    const size_t n_leaves = (size_t)(pw / 4);                // 4^D
    size_t o_dense = align_up(o_chalf + (fast ? (size_t)P * n_leaves * 16 : 0), 256);
    size_t o_part = align_up(o_dense + (size_t)P * n_dense * 4, 256);
    size_t total = align_up(o_part + (size_t)P * n_part * 4 + 4, 256);

    std::vector<unsigned char> blob(total, 0);
    int8_t *h_mip = (int8_t *)(blob.data() + o_mip);
    for (size_t i = 0; i < PN; i++) h_mip[i] = (int8_t)mip[i];
    memcpy(blob.data() + o_prob, prob4, PN * 16);
    if (fast)
        for (int p = 0; p < P; p++)
            complete_tree(mip + (size_t)p * N, prob4 + 4 * (size_t)p * N, D,
                          blob.data() + o_cshift + p * cstride,
                          (float *)(blob.data() + o_crow) + (size_t)p * ctot * 4);
    if (fast)                                                // the leaves again, two floats per half (GmgDevModel::chalf)
        for (int p = 0; p < P; p++) {
            const float *leaf = (const float *)(blob.data() + o_crow) + ((size_t)p * ctot + n_internal) * 4;
            float *half = (float *)(blob.data() + o_chalf) + (size_t)p * n_leaves * 4;
            for (size_t r = 0; r < n_leaves; r++)
                for (int h = 0; h < 2; h++) {
                    half[(h * n_leaves + r) * 2] = leaf[4 * r + 2 * h];
                    half[(h * n_leaves + r) * 2 + 1] = leaf[4 * r + 2 * h + 1];
                }
        }
    if (dense)
        for (int p = 0; p < P; p++) {
            const int16_t *pm = mip + (size_t)p * N;
            const float *pp = prob4 + 4 * (size_t)p * N;
            float *full = (float *)(blob.data() + o_dense) + (size_t)p * n_dense;
            float *part = (float *)(blob.data() + o_part) + (size_t)p * n_part;
            for (uint32_t idx = 0; idx < n_dense; idx++) full[idx] = dense_entry(pm, pp, W, D, idx);
            size_t o = 0;
            for (int j = 0; j < W - 1; j++) {
                uint32_t cnt = 1u << (2 * (j + 1));
                for (uint32_t idx = 0; idx < cnt; idx++) part[o + idx] = dense_part_entry(pm, pp, W, D, j, idx);
                o += cnt;
            }
        }

    gmg_model *m = new (std::nothrow) gmg_model();
    if (!m) return gmg_set_error(GMG_ENOMEM, "gmg_model_upload: out of host memory");
    hipError_t e = hipMalloc(&m->d_blob, total);
    if (e != hipSuccess) { delete m; return gmg_set_error(GMG_ENOMEM, "gmg_model_upload: hipMalloc(%zu): %s", total, hipGetErrorString(e)); }
    e = hipMemcpy(m->d_blob, blob.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(m->d_blob); delete m; return gmg_set_error(GMG_EHIP, "gmg_model_upload: hipMemcpy: %s", hipGetErrorString(e)); }
    m->blob_bytes = total;
    m->min_exp = 255;
    m->max_exp = 0;
    m->odd_values = 0;
    for (size_t i = 0; i < PN * 4; i++) {
        uint32_t b;
        memcpy(&b, &prob4[i], 4);
        const uint32_t ex = (b >> 23) & 0xffu;
        // (a slot without a node -- mut_info_pos -2 -- is never read: a descent that meets it stops at the parent.  A model that comes
        // from a file has zeros there, one that was just trained whatever the training left: probabilities, not logarithms.  Counting
        // those made every freshly trained table look "odd" and sent it to the sequential kernels: 13.7 instead of 10.2 ms per 1M reads)
        if (mip[i >> 2] == -2) continue;
        if ((b << 1) == 0) continue;                    // +-0: adds nothing
        if ((b >> 31) == 0 || ex == 0 || ex == 255) m->odd_values = 1;      // positive, denormal, infinity / NaN
        if ((int)ex < m->min_exp) m->min_exp = (int)ex;
        if ((int)ex > m->max_exp) m->max_exp = (int)ex;
    }
    unsigned char *d = (unsigned char *)m->d_blob;
    m->dev.W = W; m->dev.D = D; m->dev.P = P; m->dev.N = N;
    m->dev.mip = (const int8_t *)(d + o_mip);
    m->dev.prob = (const float *)(d + o_prob);
    m->dev.cshift = fast ? (const uint8_t *)(d + o_cshift) : nullptr;
    m->dev.crow = fast ? (const float *)(d + o_crow) : nullptr;
    m->dev.chalf = fast ? (const float *)(d + o_chalf) : nullptr;
    m->dev.cstride = (int)cstride;
    m->dev.ctot = (int)ctot;
    m->dev.has_fast = fast;
    m->dev.dense = dense ? (const float *)(d + o_dense) : nullptr;
    m->dev.dense_part = dense ? (const float *)(d + o_part) : nullptr;
    m->dev.n_dense_part = (int)n_part;
    m->dev.has_dense = dense;
    *out = m;
    return GMG_OK;
}

extern "C" int gmg_model_free(gmg_model *m)
{
    if (!m) return GMG_OK;
    if (m->d_blob) (void)hipFree(m->d_blob);
    delete m;
    return GMG_OK;
}

extern "C" int gmg_model_info(const gmg_model *m, int *W, int *D, int *P, int *N)
{
    if (!m) return gmg_set_error(GMG_EINVAL, "gmg_model_info: NULL model");
    if (W) *W = m->dev.W;
    if (D) *D = m->dev.D;
    if (P) *P = m->dev.P;
    if (N) *N = m->dev.N;
    return GMG_OK;
}

// A set of width-3 null models side by side (glimmer-mg -c: one per GC value the classes of a batch produce)
extern "C" int gmg_null_set_upload(const gmg_model *const *models, int n, gmg_null_set **out)
{
    int rc = require_init("gmg_null_set_upload");
    if (rc) return rc;
    if (!models || n < 1 || !out) return gmg_set_error(GMG_EINVAL, "gmg_null_set_upload: bad argument");
    for (int i = 0; i < n; i++)
        if (!models[i] || !models[i]->dev.has_dense || models[i]->dev.W != 3 || models[i]->dev.P != 3)
            return gmg_set_error(GMG_EBADMODEL, "gmg_null_set_upload: model %d is not a (3,2,3) Build_Indep_WO_Stops model", i);
    gmg_null_set *ns = new (std::nothrow) gmg_null_set();
    if (!ns) return gmg_set_error(GMG_ENOMEM, "gmg_null_set_upload: out of host memory");
    ns->d_tab = nullptr;
    ns->n = n;
    ns->min_exp = 255; ns->max_exp = 0; ns->odd_values = 0;
    for (int i = 0; i < n; i++) {
        if (models[i]->min_exp < ns->min_exp) ns->min_exp = models[i]->min_exp;
        if (models[i]->max_exp > ns->max_exp) ns->max_exp = models[i]->max_exp;
        ns->odd_values |= models[i]->odd_values;
    }
    hipError_t e = hipMalloc((void **)&ns->d_tab, (size_t)n * 252 * sizeof(float));
    for (int i = 0; i < n && e == hipSuccess; i++) {
        e = hipMemcpy(ns->d_tab + (size_t)i * 252, models[i]->dev.dense, 192 * sizeof(float), hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(ns->d_tab + (size_t)i * 252 + 192, models[i]->dev.dense_part, 60 * sizeof(float), hipMemcpyDeviceToDevice);
    }
    if (e != hipSuccess) { gmg_null_set_free(ns); return gmg_set_error(GMG_EHIP, "gmg_null_set_upload: %s", hipGetErrorString(e)); }
    *out = ns;
    return GMG_OK;
}

// The same set straight from HOST tables: n models of shape (3,2,3), mip[i][3][21], prob4[i][3][21][4] as Build_Indep_WO_Stops
// fills them (icm.cc:65-216).  The direct-lookup tables of all models are made on the host and go up in one copy -- a batch of
// glimmer-mg's classification mode can need one null model per read (the mean GC of a read's classes takes many values).
extern "C" int gmg_null_set_from_tables(const int16_t *mip, const float *prob4, int n, gmg_null_set **out)
{
    int rc = require_init("gmg_null_set_from_tables");
    if (rc) return rc;
    if (!mip || !prob4 || n < 1 || !out) return gmg_set_error(GMG_EINVAL, "gmg_null_set_from_tables: bad argument");
    for (size_t i = 0; i < (size_t)n * 63; i++)
        if (mip[i] < -2 || mip[i] > 2)
            return gmg_set_error(GMG_EBADMODEL, "gmg_null_set_from_tables: mut_info_pos %d in model %zu outside [-2,2]", (int)mip[i], i / 63);
    gmg_null_set *ns = new (std::nothrow) gmg_null_set();
    if (!ns) return gmg_set_error(GMG_ENOMEM, "gmg_null_set_from_tables: out of host memory");
    ns->d_tab = nullptr;
    ns->n = n;
    ns->min_exp = 255; ns->max_exp = 0; ns->odd_values = 0;
    std::vector<float> tab((size_t)n * 252);
    for (int i = 0; i < n; i++)
        for (int p = 0; p < 3; p++) {
            const int16_t *pm = mip + ((size_t)i * 3 + p) * 21;
            const float *pp = prob4 + ((size_t)i * 3 + p) * 21 * 4;
            float *full = tab.data() + (size_t)i * 252 + p * 64, *part = tab.data() + (size_t)i * 252 + 192 + p * 20;
            for (uint32_t idx = 0; idx < 64; idx++) full[idx] = dense_entry(pm, pp, 3, 2, idx);
            size_t o = 0;
            for (int j = 0; j < 2; j++) {
                const uint32_t cnt = 1u << (2 * (j + 1));
                for (uint32_t idx = 0; idx < cnt; idx++) part[o + idx] = dense_part_entry(pm, pp, 3, 2, j, idx);
                o += cnt;
            }
            for (int k = 0; k < 21 * 4; k++) {          // the exponent range of the values, as gmg_model_upload records it
                uint32_t b;
                memcpy(&b, pp + k, 4);
                const uint32_t ex = (b >> 23) & 255u;
                if ((b << 1) == 0) continue;
                if ((b >> 31) == 0 || ex == 0 || ex == 255) ns->odd_values = 1;
                if ((int)ex < ns->min_exp) ns->min_exp = (int)ex;
                if ((int)ex > ns->max_exp) ns->max_exp = (int)ex;
            }
        }
    hipError_t e = hipMalloc((void **)&ns->d_tab, tab.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(ns->d_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { gmg_null_set_free(ns); return gmg_set_error(GMG_EHIP, "gmg_null_set_from_tables: %s", hipGetErrorString(e)); }
    *out = ns;
    return GMG_OK;
}

extern "C" int gmg_null_set_free(gmg_null_set *ns)
{
    if (!ns) return GMG_OK;
    if (ns->d_tab) (void)hipFree(ns->d_tab);
    delete ns;
    return GMG_OK;
}

// ---------------------------------------------------------------------------
// reads
// ---------------------------------------------------------------------------

static int finish_reads(gmg_reads *r, const uint64_t *h_off /* may be NULL */)
{
    r->n_tiles = (r->total_bases + GMG_TILE - 1) / GMG_TILE;
    hipError_t e = hipMalloc((void **)&r->d_tile_read, (r->n_tiles + 1) * sizeof(uint32_t));
    if (e != hipSuccess) return gmg_set_error(GMG_ENOMEM, "gmg_reads: hipMalloc tile table: %s", hipGetErrorString(e));
    int rc = gmg_launch_tile_read(r->d_off, r->n_reads, r->n_tiles, r->d_tile_read, 0);
    if (rc) return rc;
    GMG_HIP(hipStreamSynchronize(0));
    r->uniform_len = 0;
    if (h_off && r->n_reads > 0) {
        uint64_t L = h_off[1] - h_off[0];
        bool uni = L > 0 && L < (1u << 30);
        for (uint64_t i = 1; uni && i < r->n_reads; i++) uni = (h_off[i + 1] - h_off[i] == L);
        if (uni) r->uniform_len = (int)L;
    }
    r->max_len = 0;
    r->n_over_512 = 0;
    r->min_len = r->n_reads ? ~0ull : 0;
    for (uint64_t i = 0; h_off && i < r->n_reads; i++) {
        const uint64_t len = h_off[i + 1] - h_off[i];
        if (len > r->max_len) r->max_len = len;
        if (len > 512) r->n_over_512++;
        if (len < r->min_len) r->min_len = len;
    }
    return GMG_OK;
}

// Packed reads always live in a library-owned buffer with GMG_GUARD_WORDS zero words on both
// sides, so that window loads around the first and last bases of the job stay inside it.
static int alloc_packed(gmg_reads *r, const uint32_t *src, hipMemcpyKind kind)
{
    uint64_t data_words = (r->total_bases + 15) / 16;
    r->n_words = data_words + GMG_GUARD_WORDS;        // words addressable from d_packed upwards
    uint32_t *alloc = nullptr;
    hipError_t e = hipMalloc((void **)&alloc, (data_words + 2 * GMG_GUARD_WORDS) * 4);
    if (e != hipSuccess) return gmg_set_error(GMG_ENOMEM, "gmg_reads: hipMalloc packed: %s", hipGetErrorString(e));
    r->d_packed_alloc = alloc;
    r->d_packed = alloc + GMG_GUARD_WORDS;
    GMG_HIP(hipMemset(alloc, 0, (data_words + 2 * GMG_GUARD_WORDS) * 4));
    if (data_words && src) GMG_HIP(hipMemcpy(alloc + GMG_GUARD_WORDS, src, data_words * 4, kind));
    return GMG_OK;
}

extern "C" int gmg_reads_upload(const uint32_t *packed, const uint64_t *off, uint64_t n_reads, gmg_reads **out)
{
    int rc = require_init("gmg_reads_upload");
    if (rc) return rc;
    if (!off || !out || n_reads >= 0xffffffffull) return gmg_set_error(GMG_EINVAL, "gmg_reads_upload: bad argument");
    for (uint64_t i = 0; i < n_reads; i++)
        if (off[i + 1] < off[i] || off[i + 1] - off[i] > 0x7fffffffull)
            return gmg_set_error(GMG_EINVAL, "gmg_reads_upload: base_offsets not monotone at read %llu", (unsigned long long)i);
    if (off[0] != 0) return gmg_set_error(GMG_EINVAL, "gmg_reads_upload: base_offsets[0] must be 0");
    uint64_t total = off[n_reads];
    if (total && !packed) return gmg_set_error(GMG_EINVAL, "gmg_reads_upload: NULL packed reads");
    gmg_reads *r = new (std::nothrow) gmg_reads();
    if (!r) return gmg_set_error(GMG_ENOMEM, "gmg_reads_upload: out of host memory");
    memset(r, 0, sizeof *r);
    r->n_reads = n_reads;
    r->total_bases = total;
    r->owns_off = 1;
    rc = alloc_packed(r, packed, hipMemcpyHostToDevice);
    if (rc) { gmg_reads_free(r); return rc; }
    uint64_t *d_off = nullptr;
    hipError_t e = hipMalloc((void **)&d_off, (n_reads + 1) * 8);
    if (e != hipSuccess) { gmg_reads_free(r); return gmg_set_error(GMG_ENOMEM, "gmg_reads_upload: hipMalloc: %s", hipGetErrorString(e)); }
    r->d_off = d_off;
    e = hipMemcpy(d_off, off, (n_reads + 1) * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) { gmg_reads_free(r); return gmg_set_error(GMG_EHIP, "gmg_reads_upload: copy: %s", hipGetErrorString(e)); }
    rc = finish_reads(r, off);
    if (rc) { gmg_reads_free(r); return rc; }
    *out = r;
    return GMG_OK;
}

extern "C" int gmg_reads_wrap_device(const uint32_t *d_packed, const uint64_t *d_off, uint64_t n_reads,
                                     uint64_t total_bases, gmg_reads **out)
{
    int rc = require_init("gmg_reads_wrap_device");
    if (rc) return rc;
    if (!d_off || !out || (total_bases && !d_packed) || n_reads >= 0xffffffffull)
        return gmg_set_error(GMG_EINVAL, "gmg_reads_wrap_device: bad argument");
    gmg_reads *r = new (std::nothrow) gmg_reads();
    if (!r) return gmg_set_error(GMG_ENOMEM, "gmg_reads_wrap_device: out of host memory");
    memset(r, 0, sizeof *r);
    r->d_off = d_off;
    r->n_reads = n_reads;
    r->total_bases = total_bases;
    r->owns_off = 0;
    rc = alloc_packed(r, d_packed, hipMemcpyDeviceToDevice);   // one on-device copy into a guarded buffer
    if (rc) { gmg_reads_free(r); return rc; }
    // uniform read length is detected from the offsets (a handful of bytes per read, read back once)
    std::vector<uint64_t> h_off(n_reads + 1);
    hipError_t e = hipMemcpy(h_off.data(), d_off, (n_reads + 1) * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { gmg_reads_free(r); return gmg_set_error(GMG_EHIP, "gmg_reads_wrap_device: %s", hipGetErrorString(e)); }
    if (h_off[0] != 0 || h_off[n_reads] != total_bases) {
        gmg_reads_free(r);
        return gmg_set_error(GMG_EINVAL, "gmg_reads_wrap_device: base_offsets do not span [0, total_bases]");
    }
    for (uint64_t i = 0; i < n_reads; i++)                  // the same checks as gmg_reads_upload: the kernels trust the offsets
        if (h_off[i + 1] < h_off[i] || h_off[i + 1] - h_off[i] > 0x7fffffffull) {
            gmg_reads_free(r);
            return gmg_set_error(GMG_EINVAL, "gmg_reads_wrap_device: base_offsets not monotone at read %llu", (unsigned long long)i);
        }
    rc = finish_reads(r, h_off.data());
    if (rc) { gmg_reads_free(r); return rc; }
    *out = r;
    return GMG_OK;
}

// ---- gmg_reads_select: everything on the device (a chunk of glimmer-mg's classification mode gathers all of its reads into
// visiting order: 1 M reads per call) -- lengths by index, offsets by a scan, the gather itself through the new batch's tile
// table, the batch's length statistics by a reduction; what crosses PCIe is the index list and 48 bytes back.
// stats: [0] min length, [1] max length, [2] reads over 512 bases, [3] unused, [4] first bad index
__global__ __launch_bounds__(256) void k_sel_len(const uint64_t *src_off, const uint64_t *idx, uint64_t n, uint64_t n_src, uint64_t *len,
                                                 unsigned long long *stats)
{
    // few blocks, a grid-stride loop: the four results of a WAVE go to the counters with one atomic each (every atomic on one
    // address costs ~10 ns at the L2: one per wave of a million reads was 0.36 ms)
    unsigned long long mn = ~0ull, mx = 0, over = 0, bad = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t l = 0;
        if (i < n) {
            const uint64_t r = idx[i];
            if (r >= n_src) bad = i < bad ? i : bad;
            else l = src_off[r + 1] - src_off[r];
            mn = l < mn ? l : mn;
            mx = l > mx ? l : mx;
            over += l > 512;
        }
        len[i] = l;                                     // len[n] = 0: the scan's last output is the total
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = __shfl_xor(mn, o), b = __shfl_xor(mx, o), c = __shfl_xor(bad, o);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
        bad = c < bad ? c : bad;
        over += __shfl_xor(over, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (mn != ~0ull) atomicMin(&stats[0], mn);
        if (mx) atomicMax(&stats[1], mx);
        if (over) atomicAdd(&stats[2], over);
        if (bad != ~0ull) atomicMin(&stats[4], bad);
    }
}

// One wave per 1,024-base tile of the NEW batch (64 lanes x one 16-base word): the reads that overlap the tile -- the first one
// from the tile table, their ends and source positions fetched ONCE by the wave's first lanes into LDS -- then a lane builds its
// word from one 32-base window of the source per read that overlaps it (mostly one, two at a read boundary): no global load
// depends on another one except through LDS.
#define SEL_R 32                                        // reads of a tile handled through LDS; further ones (tiles of many tiny reads) by the lane itself
__global__ __launch_bounds__(256) void k_reads_select(const uint32_t *src, const uint64_t *src_off, const uint64_t *idx, const uint64_t *new_off,
                                                      const uint32_t *tile_read, uint64_t n, uint64_t total, uint32_t *dst)
{
    __shared__ uint64_t s_end[4][SEL_R], s_src[4][SEL_R];     // per wave: end of read r0 + k in the new batch; source position of its base 0 minus its new begin
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t n_tiles = (total + GMG_TILE - 1) / GMG_TILE;
    for (uint64_t t = (uint64_t)blockIdx.x * 4 + wv; t < n_tiles; t += (uint64_t)gridDim.x * 4) {
        const uint64_t r0 = tile_read[t];
        if (lane < SEL_R) {
            const uint64_t r = r0 + lane < n ? r0 + lane : n - 1;
            const uint64_t b = new_off[r], e = r0 + lane < n ? new_off[r + 1] : ~0ull;
            s_end[wv][lane] = e;
            s_src[wv][lane] = src_off[idx[r]] - b;     // source index of new base g of this read = s_src + g (mod 2^64)
        }
        __builtin_amdgcn_wave_barrier();
        const uint64_t g0 = t * GMG_TILE + 16 * (uint64_t)lane;
        if (g0 < total) {
            const uint64_t g1 = g0 + 16 < total ? g0 + 16 : total;
            uint32_t k = 0, out = 0;
            uint64_t pos = g0;
            while (pos < g1) {
                while (k < SEL_R && s_end[wv][k] <= pos) k++;          // the read that holds base pos (empty reads are stepped over)
                if (k >= SEL_R) break;
                const uint64_t e = s_end[wv][k] < g1 ? s_end[wv][k] : g1;
                const uint64_t sg = s_src[wv][k] + pos;
                const uint64_t w0 = sg >> 4;
                const unsigned sh = 2u * (unsigned)(sg & 15);
                const uint64_t x = ((uint64_t)src[w0] | ((uint64_t)src[w0 + 1] << 32)) >> sh;      // bases pos .. pos + 15 of that read
                const unsigned nb = (unsigned)(e - pos);                                            // 1 .. 16 of them count
                out |= ((uint32_t)x & (nb >= 16 ? 0xffffffffu : ((1u << (2 * nb)) - 1u))) << (2u * (unsigned)(pos - g0));
                pos = e;
            }
            if (pos < g1) {                             // beyond the reads in LDS: base by base
                uint64_t r = r0 + SEL_R - 1;
                for (; pos < g1; pos++) {
                    while (new_off[r + 1] <= pos) r++;
                    const uint64_t sg = src_off[idx[r]] + (pos - new_off[r]);
                    out |= ((src[sg >> 4] >> (2u * (unsigned)(sg & 15))) & 3u) << (2u * (unsigned)(pos - g0));
                }
            }
            dst[t * (GMG_TILE / 16) + lane] = out;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int gmg_reads_select(const gmg_reads *reads, const uint64_t *idx, uint64_t n, gmg_reads **out)
{
    { int rc_enter = gmg_enter("gmg_reads_select"); if (rc_enter) return rc_enter; }
    if (!reads || (!idx && n) || !out || n >= 0x7ffffffeull) return gmg_set_error(GMG_EINVAL, "gmg_reads_select: bad argument");
    gmg_reads *r = new (std::nothrow) gmg_reads();
    if (!r) return gmg_set_error(GMG_ENOMEM, "gmg_reads_select: out of host memory");
    memset(r, 0, sizeof *r);
    r->n_reads = n;
    r->owns_off = 1;
    uint64_t *d_idx = nullptr, *d_len = nullptr, *d_off = nullptr;
    unsigned long long *d_stats = nullptr;
    void *d_tmp = nullptr;
    auto fail = [&](int rc) {
        (void)hipStreamSynchronize(0);                  // (kernels already queued may still use the blocks that go back to the cache)
        if (d_idx) gmg_pool_release(d_idx);
        if (d_len) gmg_pool_release(d_len);
        if (d_stats) gmg_pool_release(d_stats);
        if (d_tmp) gmg_pool_release(d_tmp);
        gmg_reads_free(r);
        return rc;
    };
#define SEL_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(gmg_set_error(e_ == hipErrorOutOfMemory ? GMG_ENOMEM : GMG_EHIP, "gmg_reads_select: %s", hipGetErrorString(e_))); } while (0)
    SEL_TRY(gmg_pool_alloc((void **)&d_idx, (n ? n : 1) * 8));
    SEL_TRY(gmg_pool_alloc((void **)&d_len, (n + 1) * 8));
    SEL_TRY(gmg_pool_alloc((void **)&d_off, (n + 1) * 8));
    r->d_off = d_off;
    SEL_TRY(gmg_pool_alloc((void **)&d_stats, 8 * sizeof(unsigned long long)));
    const unsigned long long stats0[6] = {~0ull, 0, 0, 0, ~0ull, 0};
    unsigned long long stats[6];
    hipStream_t s = 0;
    SEL_TRY(hipMemcpyAsync(d_stats, stats0, sizeof stats0, hipMemcpyHostToDevice, s));
    if (n) SEL_TRY(hipMemcpyAsync(d_idx, idx, n * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sel_len, dim3((unsigned)((n + 256) / 256 < 512 ? (n + 256) / 256 : 512)), dim3(256), 0, s, reads->d_off, d_idx, n, reads->n_reads, d_len, d_stats);
    SEL_TRY((gmg_scan_excl<uint64_t, uint64_t>(d_len, d_off, n + 1, s)));
    SEL_TRY(hipMemcpyAsync(&stats[5], d_off + n, 8, hipMemcpyDeviceToHost, s));
    SEL_TRY(hipMemcpyAsync(stats, d_stats, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    SEL_TRY(hipStreamSynchronize(s));
    if (stats[4] != ~0ull)
        return fail(gmg_set_error(GMG_ERANGE, "gmg_reads_select: entry %llu names a read beyond the batch's %llu", stats[4], (unsigned long long)reads->n_reads));
    r->total_bases = stats[5];
    // the packed words (guards zeroed; the gather writes every data word), the tile table, the gather
    const uint64_t data_words = (r->total_bases + 15) / 16;
    r->n_words = data_words + GMG_GUARD_WORDS;
    uint32_t *alloc = nullptr;
    SEL_TRY(gmg_pool_alloc((void **)&alloc, (data_words + 2 * GMG_GUARD_WORDS) * 4));
    r->d_packed_alloc = alloc;
    r->d_packed = alloc + GMG_GUARD_WORDS;
    SEL_TRY(hipMemsetAsync(alloc, 0, GMG_GUARD_WORDS * 4, s));
    SEL_TRY(hipMemsetAsync(alloc + GMG_GUARD_WORDS + data_words, 0, GMG_GUARD_WORDS * 4, s));
    r->n_tiles = (r->total_bases + GMG_TILE - 1) / GMG_TILE;
    SEL_TRY(gmg_pool_alloc((void **)&r->d_tile_read, (r->n_tiles + 1) * sizeof(uint32_t)));
    {
        const int rc = gmg_launch_tile_read(r->d_off, r->n_reads, r->n_tiles, r->d_tile_read, s);
        if (rc) return fail(rc);
    }
    if (data_words) {
        const uint64_t blocks = (r->n_tiles + 3) / 4;      // one wave per tile
        hipLaunchKernelGGL(k_reads_select, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, reads->d_packed, reads->d_off, d_idx,
                           d_off, r->d_tile_read, n, r->total_bases, alloc + GMG_GUARD_WORDS);
        SEL_TRY(hipGetLastError());
    }
    SEL_TRY(hipStreamSynchronize(s));
#undef SEL_TRY
    gmg_pool_release(d_idx);
    gmg_pool_release(d_len);
    gmg_pool_release(d_stats);
    if (d_tmp) gmg_pool_release(d_tmp);
    r->min_len = n ? stats[0] : 0;
    r->max_len = stats[1];
    r->n_over_512 = stats[2];
    r->uniform_len = (n && stats[0] == stats[1] && stats[0] > 0 && stats[0] < (1u << 30)) ? (int)stats[0] : 0;
    *out = r;
    return GMG_OK;
}

extern "C" int gmg_reads_free(gmg_reads *r)
{
    if (!r) return GMG_OK;
    // (gmg_pool_release: a block of the library's cache goes back to it, anything else is hipFree'd.  A cached block is handed out
    // again at once: kernels queued on a caller's stream may still read this batch -- hipFree waited for them, the cache does not)
    (void)hipDeviceSynchronize();
    if (r->d_packed_alloc) gmg_pool_release(r->d_packed_alloc);
    if (r->owns_off && r->d_off) gmg_pool_release((void *)r->d_off);
    if (r->d_tile_read) gmg_pool_release(r->d_tile_read);
    delete r;
    return GMG_OK;
}

extern "C" int gmg_reads_info(const gmg_reads *r, uint64_t *n_reads, uint64_t *total_bases)
{
    if (!r) return gmg_set_error(GMG_EINVAL, "gmg_reads_info: NULL reads");
    if (n_reads) *n_reads = r->n_reads;
    if (total_bases) *total_bases = r->total_bases;
    return GMG_OK;
}

extern "C" int gmg_reads_download(const gmg_reads *r, uint32_t *packed, uint64_t *off)
{
    { int rc = gmg_enter("gmg_reads_download"); if (rc) return rc; }
    if (!r || !off || (r->total_bases && !packed)) return gmg_set_error(GMG_EINVAL, "gmg_reads_download: NULL argument");
    if (r->total_bases)
        GMG_HIP(hipMemcpy(packed, r->d_packed, gmg_packed_words(r->total_bases) * 4, hipMemcpyDeviceToHost));
    GMG_HIP(hipMemcpy(off, r->d_off, (r->n_reads + 1) * 8, hipMemcpyDeviceToHost));
    return GMG_OK;
}

// ---------------------------------------------------------------------------
// one string at a time (the ICM_t methods): persistent staging, no allocation per call
// ---------------------------------------------------------------------------
struct gmg_single {
    gmg_reads reads;
    gmg_segments segs;
    uint64_t cap_bases;          // capacity of everything below
    unsigned char *h_in;         // page-locked: [packed words + guard][off[2]][segment][out_off[2]]
    unsigned char *d_in;         // the same block on the device, behind GMG_GUARD_WORDS zero words
    double *h_out, *d_out;       // results (page-locked / device), cap_bases + 16 doubles
    uint32_t *d_tile_read;       // zeros: every tile lies in read 0
    bool in_flight;              // a staged copy may still read h_in
};

static void single_release(gmg_single *st)
{
    if (st->h_in) (void)hipHostFree(st->h_in);
    if (st->d_in) (void)hipFree(st->d_in);
    if (st->h_out) (void)hipHostFree(st->h_out);
    if (st->d_out) (void)hipFree(st->d_out);
    if (st->d_tile_read) (void)hipFree(st->d_tile_read);
    st->h_in = st->d_in = nullptr;
    st->h_out = st->d_out = nullptr;
    st->d_tile_read = nullptr;
    st->cap_bases = 0;
}

static size_t single_words(uint64_t bases) { return (size_t)((bases + 15) / 16 + GMG_GUARD_WORDS); }   // data + trailing guard

static int single_reserve(gmg_single *st, uint64_t n)
{
    if (n <= st->cap_bases && st->h_in) return GMG_OK;
    single_release(st);
    uint64_t cap = 4096;
    while (cap < n) cap *= 2;
    const size_t in_bytes = single_words(cap) * 4 + 16 + sizeof(gmg_segment) + 16;
    hipError_t e = hipHostMalloc((void **)&st->h_in, in_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&st->d_in, GMG_GUARD_WORDS * 4 + in_bytes);
    if (e == hipSuccess) e = hipMemset(st->d_in, 0, GMG_GUARD_WORDS * 4 + in_bytes);
    if (e == hipSuccess) e = hipHostMalloc((void **)&st->h_out, (cap + 16) * sizeof(double), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&st->d_out, (cap + 16) * sizeof(double));
    const size_t tiles = (size_t)(cap / GMG_TILE + 2);
    if (e == hipSuccess) e = hipMalloc((void **)&st->d_tile_read, tiles * 4);
    if (e == hipSuccess) e = hipMemset(st->d_tile_read, 0, tiles * 4);
    if (e != hipSuccess) { single_release(st); return gmg_set_error(GMG_ENOMEM, "gmg_single: %s", hipGetErrorString(e)); }
    st->cap_bases = cap;
    return GMG_OK;
}

extern "C" int gmg_single_create(gmg_single **out)
{
    int rc = require_init("gmg_single_create");
    if (rc) return rc;
    if (!out) return gmg_set_error(GMG_EINVAL, "gmg_single_create: NULL argument");
    gmg_single *st = new (std::nothrow) gmg_single();
    if (!st) return gmg_set_error(GMG_ENOMEM, "gmg_single_create: out of host memory");
    memset(st, 0, sizeof *st);
    *out = st;
    return GMG_OK;
}

extern "C" int gmg_single_free(gmg_single *st)
{
    if (!st) return GMG_OK;
    single_release(st);
    delete st;
    return GMG_OK;
}

extern "C" int gmg_single_stage(gmg_single *st, const char *ascii, uint64_t n, int orient, const gmg_reads **reads,
                                const gmg_segments **segs, double **d_out)
{
    int rc = require_init("gmg_single_stage");
    if (rc) return rc;
    if (!st || (!ascii && n) || !reads || !segs || !d_out || orient < 0 || orient > GMG_REVCOMP || n > 0x7fffffffull)
        return gmg_set_error(GMG_EINVAL, "gmg_single_stage: bad argument");
    if (st->in_flight) GMG_HIP(hipStreamSynchronize(0));            // (a stage without a fetch: the last copy may still read h_in)
    if ((rc = single_reserve(st, n)) != GMG_OK) return rc;
    // host block: the packed words with the zero guard behind them, then the read's offsets, its segment, the output offsets
    const size_t words = single_words(n);
    uint32_t *h_packed = (uint32_t *)st->h_in;
    memset(h_packed, 0, words * 4);
    if (n && (rc = gmg_pack_bases(ascii, n, 0, h_packed)) != GMG_OK) return rc;
    unsigned char *h_tail = st->h_in + single_words(st->cap_bases) * 4;
    uint64_t off[2] = {0, n};
    gmg_segment sg = {0, 0, (uint32_t)n, (uint32_t)orient};
    memcpy(h_tail, off, 16);
    memcpy(h_tail + 16, &sg, sizeof sg);
    memcpy(h_tail + 16 + sizeof sg, off, 16);
    unsigned char *d_packed = st->d_in + GMG_GUARD_WORDS * 4, *d_tail = d_packed + single_words(st->cap_bases) * 4;
    GMG_HIP(hipMemcpyAsync(d_packed, h_packed, words * 4, hipMemcpyHostToDevice, 0));
    GMG_HIP(hipMemcpyAsync(d_tail, h_tail, 16 + sizeof sg + 16, hipMemcpyHostToDevice, 0));
    gmg_reads &r = st->reads;
    memset(&r, 0, sizeof r);
    r.d_packed = (const uint32_t *)d_packed;
    r.d_off = (const uint64_t *)d_tail;
    r.d_tile_read = st->d_tile_read;
    r.n_reads = 1;
    r.total_bases = n;
    r.n_tiles = (n + GMG_TILE - 1) / GMG_TILE;
    r.n_words = words;
    r.d_packed_alloc = st->d_in;
    r.uniform_len = n > 0 && n < (1u << 30) ? (int)n : 0;
    r.max_len = r.min_len = n;
    r.n_over_512 = n > 512;
    gmg_segments &g = st->segs;
    g.d_segs = (gmg_segment *)(d_tail + 16);
    g.d_out_off = (uint64_t *)(d_tail + 16 + sizeof sg);
    g.n = 1;
    g.total_len = n;
    st->in_flight = true;
    *reads = &r;
    *segs = &g;
    *d_out = st->d_out;
    return GMG_OK;
}

extern "C" int gmg_single_fetch(gmg_single *st, double *dst, size_t n)
{
    int rc = require_init("gmg_single_fetch");
    if (rc) return rc;
    if (!st || (!dst && n) || n > st->cap_bases + 16) return gmg_set_error(GMG_EINVAL, "gmg_single_fetch: bad argument");
    if (n == 0) return GMG_OK;
    GMG_HIP(hipMemcpyAsync(st->h_out, st->d_out, n * sizeof(double), hipMemcpyDeviceToHost, 0));
    GMG_HIP(hipStreamSynchronize(0));
    st->in_flight = false;
    memcpy(dst, st->h_out, n * sizeof(double));
    return GMG_OK;
}

// ICM_t::Full_Window_Prob / Full_Window_Distrib for ONE window (src/ICM/icm.cc:512-610) on the staging of a gmg_single: the codes and the
// sub-model go up in one copy, gmg_window_distrib runs on them, prob and the four floats come back in one copy -- nothing is allocated
extern "C" int gmg_single_window(gmg_single *st, const gmg_model *m, const uint8_t *codes, int model_len, int frame, float *dist4, double *prob)
{
    int rc = require_init("gmg_single_window");
    if (rc) return rc;
    if (!st || !m || !codes || model_len < 1 || model_len > 1024) return gmg_set_error(GMG_EINVAL, "gmg_single_window: bad argument");
    if (st->in_flight) GMG_HIP(hipStreamSynchronize(0));
    if ((rc = single_reserve(st, 4096)) != GMG_OK) return rc;
    const size_t fr_off = ((size_t)model_len + 7) & ~(size_t)7;
    memcpy(st->h_in, codes, (size_t)model_len);
    const int32_t fr = frame;
    memcpy(st->h_in + fr_off, &fr, 4);
    unsigned char *d_blk = st->d_in + GMG_GUARD_WORDS * 4;
    GMG_HIP(hipMemcpyAsync(d_blk, st->h_in, fr_off + 4, hipMemcpyHostToDevice, 0));
    st->in_flight = true;
    rc = gmg_window_distrib(m, (const uint8_t *)d_blk, (const int32_t *)(d_blk + fr_off), 1, (float *)(st->d_out + 1), st->d_out, nullptr);
    if (rc) return rc;
    GMG_HIP(hipMemcpyAsync(st->h_out, st->d_out, 24, hipMemcpyDeviceToHost, 0));
    GMG_HIP(hipStreamSynchronize(0));
    st->in_flight = false;
    if (prob) memcpy(prob, st->h_out, 8);
    if (dist4) memcpy(dist4, st->h_out + 1, 16);
    // (the block is shared with gmg_single_stage, which rewrites the words of its read and GMG_GUARD_WORDS zero words behind them:
    // more than a window occupies)
    return GMG_OK;
}

// ---------------------------------------------------------------------------
// segments
// ---------------------------------------------------------------------------

extern "C" int gmg_segments_upload(const gmg_reads *reads, const gmg_segment *segs, uint64_t n,
                                   uint64_t *out_offsets, uint64_t *out_total_len, gmg_segments **out)
{
    int rc = require_init("gmg_segments_upload");
    if (rc) return rc;
    if (!reads || (!segs && n) || !out) return gmg_set_error(GMG_EINVAL, "gmg_segments_upload: NULL argument");
    // read lengths are needed for validation: fetch the offsets once
    std::vector<uint64_t> off(reads->n_reads + 1);
    GMG_HIP(hipMemcpy(off.data(), reads->d_off, off.size() * 8, hipMemcpyDeviceToHost));
    std::vector<uint64_t> pre(n + 1);
    pre[0] = 0;
    for (uint64_t i = 0; i < n; i++) {
        const gmg_segment &s = segs[i];
        if (s.read >= reads->n_reads || s.orient > GMG_REVCOMP)
            return gmg_set_error(GMG_ERANGE, "gmg_segments_upload: segment %llu: bad read %u / orient %u",
                                 (unsigned long long)i, s.read, s.orient);
        uint64_t L = off[s.read + 1] - off[s.read];
        if ((uint64_t)s.lo + s.len > L)
            return gmg_set_error(GMG_ERANGE, "gmg_segments_upload: segment %llu [%u,+%u) leaves read %u of length %llu",
                                 (unsigned long long)i, s.lo, s.len, s.read, (unsigned long long)L);
        pre[i + 1] = pre[i] + s.len;
    }
    gmg_segments *g = new (std::nothrow) gmg_segments();
    if (!g) return gmg_set_error(GMG_ENOMEM, "gmg_segments_upload: out of host memory");
    memset(g, 0, sizeof *g);
    g->n = n;
    g->total_len = pre[n];
    hipError_t e = hipMalloc((void **)&g->d_segs, (n ? n : 1) * sizeof(gmg_segment));
    if (e == hipSuccess) e = hipMalloc((void **)&g->d_out_off, (n + 1) * 8);
    if (e == hipSuccess && n) e = hipMemcpy(g->d_segs, segs, n * sizeof(gmg_segment), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->d_out_off, pre.data(), (n + 1) * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) { gmg_segments_free(g); return gmg_set_error(GMG_EHIP, "gmg_segments_upload: %s", hipGetErrorString(e)); }
    if (out_offsets) memcpy(out_offsets, pre.data(), (n + 1) * 8);
    if (out_total_len) *out_total_len = pre[n];
    *out = g;
    return GMG_OK;
}

extern "C" int gmg_segments_free(gmg_segments *s)
{
    if (!s) return GMG_OK;
    if (s->d_segs) (void)hipFree(s->d_segs);
    if (s->d_out_off) (void)hipFree(s->d_out_off);
    delete s;
    return GMG_OK;
}

// ---------------------------------------------------------------------------
// scoring entry points: argument checks, then the launchers in gmg_kernels.hip
// ---------------------------------------------------------------------------

static int check_frame(const gmg_model *m, int frame, const char *who)
{
    // src/ICM/icm.cc:367-369,875-877: periodicity 1 forces frame 0, otherwise assert(0 <= frame < periodicity)
    if (m->dev.P == 1) return GMG_OK;
    if (frame < 0 || frame >= m->dev.P)
        return gmg_set_error(GMG_EINVAL, "%s: frame %d outside [0,%d)", who, frame, m->dev.P);
    return GMG_OK;
}

extern "C" int gmg_frame_score6(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads,
                                double *d_out, void *stream)
{
    int rc = require_init("gmg_frame_score6");
    if (rc) return rc;
    if (!gene || !nul || !reads || (!d_out && reads->total_bases))
        return gmg_set_error(GMG_EINVAL, "gmg_frame_score6: NULL argument");
    // Score_All_Frames calls Frame_Score with frame 0..2, which asserts frame < periodicity (icm.cc:496)
    if (gene->dev.P < 3 || nul->dev.P < 3)
        return gmg_set_error(GMG_EBADMODEL, "gmg_frame_score6: periodicity must be >= 3 (gene %d, null %d)",
                             gene->dev.P, nul->dev.P);
    if (reads->total_bases == 0) return GMG_OK;
    return gmg_launch_frame6(gene, nul, reads, d_out, (hipStream_t)stream);
}

extern "C" int gmg_frame_score6_strided(const gmg_model *gene, const gmg_model *nul, const gmg_reads *reads,
                                        double *d_out, uint64_t row_stride, void *stream)
{
    int rc = require_init("gmg_frame_score6_strided");
    if (rc) return rc;
    if (!gene || !nul || !reads || (!d_out && reads->total_bases))
        return gmg_set_error(GMG_EINVAL, "gmg_frame_score6_strided: NULL argument");
    if (row_stride < reads->total_bases)
        return gmg_set_error(GMG_EINVAL, "gmg_frame_score6_strided: row stride %llu < total_bases %llu",
                             (unsigned long long)row_stride, (unsigned long long)reads->total_bases);
    if (gene->dev.P < 3 || nul->dev.P < 3)
        return gmg_set_error(GMG_EBADMODEL, "gmg_frame_score6_strided: periodicity must be >= 3 (gene %d, null %d)",
                             gene->dev.P, nul->dev.P);
    if (reads->total_bases == 0) return GMG_OK;
    return gmg_launch_frame6_strided(gene, nul, reads, d_out, row_stride, (hipStream_t)stream);
}

extern "C" int gmg_segment_frame_score(const gmg_model *m, const gmg_reads *reads, const gmg_segments *segs,
                                       int frame, double *d_out, void *stream)
{
    int rc = require_init("gmg_segment_frame_score");
    if (rc) return rc;
    if (!m || !reads || !segs || (!d_out && segs->total_len))
        return gmg_set_error(GMG_EINVAL, "gmg_segment_frame_score: NULL argument");
    // Frame_Score asserts the range even for periodicity 1 (src/ICM/icm.cc:496)
    if (frame < 0 || frame >= m->dev.P)
        return gmg_set_error(GMG_EINVAL, "gmg_segment_frame_score: frame %d outside [0,%d)", frame, m->dev.P);
    if (segs->total_len == 0) return GMG_OK;
    return gmg_launch_seg_frame(m, reads, segs, frame, d_out, (hipStream_t)stream);
}

extern "C" int gmg_segment_cumscore(const gmg_model *m, const gmg_reads *reads, const gmg_segments *segs,
                                    int frame0, double *d_out, void *stream)
{
    int rc = require_init("gmg_segment_cumscore");
    if (rc) return rc;
    if (!m || !reads || !segs || (!d_out && segs->total_len))
        return gmg_set_error(GMG_EINVAL, "gmg_segment_cumscore: NULL argument");
    if ((rc = check_frame(m, frame0, "gmg_segment_cumscore"))) return rc;
    if (segs->n == 0) return GMG_OK;
    return gmg_launch_seg_cum(m, reads, segs, m->dev.P == 1 ? 0 : frame0, d_out, nullptr, (hipStream_t)stream);
}

extern "C" int gmg_score_string(const gmg_model *m, const gmg_reads *reads, const gmg_segments *segs,
                                int frame0, double *d_sums, void *stream)
{
    int rc = require_init("gmg_score_string");
    if (rc) return rc;
    if (!m || !reads || !segs || (!d_sums && segs->n))
        return gmg_set_error(GMG_EINVAL, "gmg_score_string: NULL argument");
    if ((rc = check_frame(m, frame0, "gmg_score_string"))) return rc;
    if (segs->n == 0) return GMG_OK;
    return gmg_launch_seg_cum(m, reads, segs, m->dev.P == 1 ? 0 : frame0, nullptr, d_sums, (hipStream_t)stream);
}

extern "C" int gmg_segment_partial_prob(const gmg_model *m, const gmg_reads *reads, const gmg_segments *segs,
                                        int frame, double *d_out, void *stream)
{
    int rc = require_init("gmg_segment_partial_prob");
    if (rc) return rc;
    if (!m || !reads || !segs || (!d_out && segs->n))
        return gmg_set_error(GMG_EINVAL, "gmg_segment_partial_prob: NULL argument");
    if (frame < 0 || frame >= m->dev.P)
        return gmg_set_error(GMG_EINVAL, "gmg_segment_partial_prob: frame %d outside [0,%d)", frame, m->dev.P);
    if (segs->n == 0) return GMG_OK;
    return gmg_launch_seg_partial(m, reads, segs, frame, d_out, (hipStream_t)stream);
}

extern "C" int gmg_all_frame_score(const gmg_model *gene, const gmg_reads *reads, const gmg_segments *segs,
                                   const uint32_t *d_prefix_len, const int32_t *d_frame, double *d_af,
                                   void *stream)
{
    int rc = require_init("gmg_all_frame_score");
    if (rc) return rc;
    if (!gene || !reads || !segs || (segs->n && (!d_prefix_len || !d_frame || !d_af)))
        return gmg_set_error(GMG_EINVAL, "gmg_all_frame_score: NULL argument");
    if (gene->dev.P != 3 && gene->dev.P != 1)
        return gmg_set_error(GMG_EBADMODEL, "gmg_all_frame_score: periodicity must be 3 or 1");
    if (segs->n == 0) return GMG_OK;
    return gmg_launch_all_frame(gene, reads, segs, d_prefix_len, d_frame, d_af, (hipStream_t)stream);
}

extern "C" int gmg_window_distrib(const gmg_model *m, const uint8_t *d_windows, const int32_t *d_frames,
                                  uint64_t n, float *d_dist4, double *d_prob, void *stream)
{
    int rc = require_init("gmg_window_distrib");
    if (rc) return rc;
    if (!m || (n && (!d_windows || !d_frames)))
        return gmg_set_error(GMG_EINVAL, "gmg_window_distrib: NULL argument");
    if (n == 0) return GMG_OK;
    return gmg_launch_windows(m, d_windows, d_frames, n, d_dist4, d_prob, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// memory helpers
// ---------------------------------------------------------------------------

extern "C" int gmg_device_malloc(void **d_ptr, size_t bytes)
{
    int rc = require_init("gmg_device_malloc");
    if (rc) return rc;
    if (!d_ptr) return gmg_set_error(GMG_EINVAL, "gmg_device_malloc: NULL argument");
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) return gmg_set_error(GMG_ENOMEM, "gmg_device_malloc(%zu): %s", bytes, hipGetErrorString(e));
    return GMG_OK;
}

extern "C" int gmg_device_free(void *d_ptr)
{
    if (d_ptr) GMG_HIP(hipFree(d_ptr));
    return GMG_OK;
}

extern "C" int gmg_memcpy_h2d(void *d_dst, const void *src, size_t bytes, void *stream)
{
    { int rc = gmg_enter("gmg_memcpy_h2d"); if (rc) return rc; }
    GMG_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    GMG_HIP(hipStreamSynchronize((hipStream_t)stream));
    return GMG_OK;
}

extern "C" int gmg_memcpy_d2h(void *dst, const void *d_src, size_t bytes, void *stream)
{
    { int rc = gmg_enter("gmg_memcpy_d2h"); if (rc) return rc; }
    GMG_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    GMG_HIP(hipStreamSynchronize((hipStream_t)stream));
    return GMG_OK;
}

// ---------------------------------------------------------------------------------------------------
// Scratch and result buffers come from a small cache of device blocks: hipMalloc / hipFree of GB-sized
// buffers cost up to hundreds of milliseconds now and then (measured: tests/bench/bench_mg.py), far more than the
// kernels.  A released block is kept and handed to the next request it fits (size <= block <= 2 x size);
// gmg_trim_cache() gives everything back to the driver.
// ---------------------------------------------------------------------------------------------------
namespace {
struct PoolBlock { void *p; size_t bytes; bool busy; hipEvent_t pending; bool waiting; };   // waiting: free once `pending` has passed
std::mutex g_pool_mutex;
std::vector<PoolBlock> g_pool;

}  // namespace

hipError_t gmg_pool_alloc(void **out, size_t bytes)
{
    if (bytes == 0) bytes = 1;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    int best = -1;
    for (auto &b : g_pool)                              // blocks released "after the stream reaches here" (gmg_pool_release_after)
        if (b.busy && b.waiting && hipEventQuery(b.pending) == hipSuccess) { b.busy = false; b.waiting = false; }
    for (size_t i = 0; i < g_pool.size(); i++)
        if (!g_pool[i].busy && g_pool[i].bytes >= bytes && g_pool[i].bytes <= 2 * bytes + 4096 &&
            (best < 0 || g_pool[i].bytes < g_pool[best].bytes))
            best = (int)i;
    if (best >= 0) { g_pool[best].busy = true; *out = g_pool[best].p; return hipSuccess; }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {                              // make room: drop the idle blocks and try once more
        for (size_t i = 0; i < g_pool.size();)
            if (!g_pool[i].busy) { if (g_pool[i].pending) (void)hipEventDestroy(g_pool[i].pending); (void)hipFree(g_pool[i].p); g_pool.erase(g_pool.begin() + i); } else i++;
        (void)hipGetLastError();
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return e;
    }
    g_pool.push_back({p, bytes, true, nullptr, false});
    *out = p;
    return hipSuccess;
}

// The block goes back to the cache once everything queued on `s` so far has run (an asynchronous entry point's scratch).
void gmg_pool_release_after(void *p, hipStream_t s)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (auto &b : g_pool)
        if (b.p == p) {
            if (!b.pending && hipEventCreateWithFlags(&b.pending, hipEventDisableTiming) != hipSuccess) b.pending = nullptr;
            if (b.pending && hipEventRecord(b.pending, s) == hipSuccess) { b.waiting = true; return; }
            (void)hipStreamSynchronize(s);              // no event: wait here instead
            b.busy = false;
            return;
        }
}

void gmg_pool_release(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (auto &b : g_pool)
        if (b.p == p) { b.busy = false; b.waiting = false; return; }
    (void)hipFree(p);                                   // not ours
}

extern "C" int gmg_trim_cache(void)
{
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (auto &b : g_pool)
        if (b.busy && b.waiting && hipEventQuery(b.pending) == hipSuccess) { b.busy = false; b.waiting = false; }
    for (size_t i = 0; i < g_pool.size();)
        if (!g_pool[i].busy) { if (g_pool[i].pending) (void)hipEventDestroy(g_pool[i].pending); (void)hipFree(g_pool[i].p); g_pool.erase(g_pool.begin() + i); } else i++;
    gmg_ingest_trim();                                  // (the page-locked header buffers gmg_fasta_ingest keeps)
    return GMG_OK;
}


// Page-lock a host buffer the caller already owns (file bytes, result arrays): copies to and from it then run at
// PCIe speed instead of through the runtime's staging buffers.
extern "C" int gmg_host_register(void *ptr, size_t bytes)
{
    { int rc = gmg_enter("gmg_host_register"); if (rc) return rc; }
    if (!ptr || !bytes) return gmg_set_error(GMG_EINVAL, "gmg_host_register: empty buffer");
    GMG_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return GMG_OK;
}

extern "C" int gmg_host_unregister(void *ptr)
{
    if (!ptr) return GMG_OK;
    GMG_HIP(hipHostUnregister(ptr));
    return GMG_OK;
}

// Streams for callers that overlap the stages of a pipeline (ingest of the next piece of a file, scoring, copying the
// previous results back) without a HIP binding of their own.  Non-blocking: no implicit ordering with the null stream.
extern "C" int gmg_stream_create(void **stream)
{
    int rc = require_init("gmg_stream_create");
    if (rc) return rc;
    if (!stream) return gmg_set_error(GMG_EINVAL, "gmg_stream_create: NULL argument");
    hipStream_t s;
    GMG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return GMG_OK;
}

extern "C" int gmg_stream_destroy(void *stream)
{
    if (stream) GMG_HIP(hipStreamDestroy((hipStream_t)stream));
    return GMG_OK;
}
