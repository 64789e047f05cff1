/*
 * gmg_oracle.c -- CPU ORACLE (test infrastructure, never shipped in the product
 * path).  See gmg_oracle.h for the parity status and the usage rules.
 *
 * Plain-C restatement of the reference algorithm.  It deliberately keeps the
 * reference's data flow (NUL-terminated lower-case strings, per-call tree
 * descent, sequential double adds) so that every value is bit-identical to
 * what the reference computes; it is not meant to be fast.
 */
#include "gmg_oracle.h"

#include <ctype.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_ALPHA 4
#define ORC_ID_STRING_LEN 150 /* icm.hh:46 */
#define ORC_VERSION_ID 200    /* icm.hh:80 */

/* icm.hh:84  PARENT(x) = (x-1)/4 with C truncation (parent of the root is the root) */
static int orc_parent(int x) { return (x - 1) / ORC_ALPHA; }

/* ------------------------------------------------------------------------ */
/* characters                                                               */
/* ------------------------------------------------------------------------ */

/* gene.cc:1139-1175.  a/c/g/t keep their case; IUPAC codes collapse to one
 * lower-case base; everything else becomes 'c'. */
int orc_filter(int ch)
{
    switch (tolower(ch)) {
    case 'a': case 'c': case 'g': case 't': return ch;
    case 'r': case 'd':                     return 'g';
    case 'w': case 'k':                     return 't';
    default:                                return 'c';   /* y s m b h v and anything else */
    }
}

/* gene.cc:15-19 table, restated as rules: IUPAC complement with case kept,
 * a few punctuation marks map to themselves, the rest to 'n'. */
int orc_complement(int ch)
{
    static const char from[] = "acgtrykmbdhvswn";
    static const char to[]   = "tgcayrmkvhdbswn";
    const char *p;
    int lower;
    if (ch < 32 || ch > 127) return 'n';
    if (ch == ' ' || ch == '*' || ch == '-' || ch == '.' || ch == '_') return ch;
    if (!isalpha(ch)) return 'n';
    lower = tolower(ch);
    p = strchr(from, lower);
    if (p == NULL) return (ch == lower) ? 'n' : 'N';
    return (ch == lower) ? to[p - from] : toupper(to[p - from]);
}

/* icm.cc:2008-2027.  The reference exits on a bad character; the oracle
 * returns -1 (unreachable after orc_filter). */
int orc_subscript(int ch)
{
    switch (tolower(orc_filter(ch))) {
    case 'a': return 0;
    case 'c': return 1;
    case 'g': return 2;
    case 't': return 3;
    }
    return -1;
}

/* ------------------------------------------------------------------------ */
/* model construction / IO                                                  */
/* ------------------------------------------------------------------------ */

static int orc_alloc_tables(orc_model *m)
{
    size_t n = (size_t)m->periodicity * (size_t)m->num_nodes;
    m->mip = (int16_t *)calloc(n ? n : 1, sizeof(int16_t));
    m->prob = (float *)calloc(n ? 4 * n : 4, sizeof(float));
    return (m->mip && m->prob) ? 0 : -1;
}

/* icm.cc:24-44: num_nodes = (4^(D+1) - 1) / 3, zero-filled tables */
orc_model *orc_model_new(int model_len, int model_depth, int periodicity)
{
    orc_model *m = (orc_model *)calloc(1, sizeof(*m));
    int i, pw = 1;
    if (!m) return NULL;
    for (i = 0; i < model_depth + 1; i++) pw *= ORC_ALPHA;
    m->model_len = model_len;
    m->model_depth = model_depth;
    m->periodicity = periodicity;
    m->num_nodes = (pw - 1) / (ORC_ALPHA - 1);
    if (orc_alloc_tables(m) != 0) { orc_model_free(m); return NULL; }
    return m;
}

void orc_model_free(orc_model *m)
{
    if (!m) return;
    free(m->mip);
    free(m->prob);
    free(m);
}

static int32_t rd_i32(const unsigned char *p) { int32_t v; memcpy(&v, p, 4); return v; }

/* icm.cc:614-726.  Layout: 150-byte text header, 6 x int32
 * {version, id_string_len, model_len, depth, periodicity, num_nodes}, then
 * records {int32 id, 4 x float prob, int16 mip}; a record with id 0 opens the
 * next sub-model; ids skipped in between are cut nodes (mip = -2); a negative
 * id terminates the stream. */
orc_model *orc_model_from_bytes(const unsigned char *buf, size_t n, char *err, size_t errlen)
{
    orc_model *m = NULL;
    size_t off = 0;
    int32_t par[6];
    int i, period = -1, prev_node = 0;

#define ORC_FAIL(...) do { if (err) snprintf(err, errlen, __VA_ARGS__); orc_model_free(m); return NULL; } while (0)

    if (n < ORC_ID_STRING_LEN) ORC_FAIL("ERROR reading ICM header");
    off = ORC_ID_STRING_LEN;
    if (n < off + 24) ORC_FAIL("ERROR reading parameters");
    for (i = 0; i < 6; i++) par[i] = rd_i32(buf + off + 4 * i);
    off += 24;
    if (par[0] != ORC_VERSION_ID) ORC_FAIL("Bad ICM version = %d  should be %d", par[0], ORC_VERSION_ID);
    if (par[1] != ORC_ID_STRING_LEN) ORC_FAIL("Bad ID_STRING_LEN = %d  should be %d", par[1], ORC_ID_STRING_LEN);

    m = (orc_model *)calloc(1, sizeof(*m));
    if (!m) ORC_FAIL("out of memory");
    m->model_len = par[2];
    m->model_depth = par[3];
    m->periodicity = par[4];
    m->num_nodes = par[5];
    if (m->periodicity <= 0 || m->num_nodes <= 0 || m->model_len <= 0 || m->model_depth < 0)
        ORC_FAIL("ERROR:  bad ICM parameters");
    if (orc_alloc_tables(m) != 0) ORC_FAIL("out of memory");

    while (off + 4 <= n) {
        int32_t id = rd_i32(buf + off);
        size_t slot;
        off += 4;
        if (id < 0) break;
        if (id == 0) period++;
        if (period < 0 || period >= m->periodicity || id >= m->num_nodes)
            ORC_FAIL("ERROR reading icm node = %d  period = %d", id, period);
        if (off + 16 > n) ORC_FAIL("ERROR reading icm node = %d  period = %d", id, period);
        slot = (size_t)period * m->num_nodes + id;
        memcpy(m->prob + 4 * slot, buf + off, 16);
        off += 16;
        if (off + 2 > n) ORC_FAIL("ERROR reading mut_info_pos for node = %d  period = %d", id, period);
        memcpy(m->mip + slot, buf + off, 2);
        off += 2;
        /* gaps in the id sequence are cut nodes */
        if (id != 0 && prev_node != id - 1)
            for (i = prev_node + 1; i < id; i++)
                m->mip[(size_t)period * m->num_nodes + i] = -2;
        if (id == 0 && period > 0)
            for (i = prev_node + 1; i < m->num_nodes; i++)
                m->mip[(size_t)(period - 1) * m->num_nodes + i] = -2;
        prev_node = id;
    }
    if (period != m->periodicity - 1)
        ORC_FAIL("ERROR:  Too few nodes for periodicity = %d", m->periodicity);
    if (prev_node != m->num_nodes - 1)
        for (i = prev_node + 1; i < m->num_nodes; i++)
            m->mip[(size_t)period * m->num_nodes + i] = -2;
#undef ORC_FAIL
    return m;
}

orc_model *orc_model_read(const char *path, char *err, size_t errlen)
{
    FILE *fp = fopen(path, "rb");
    unsigned char *buf;
    long sz;
    orc_model *m;
    if (!fp) { if (err) snprintf(err, errlen, "ERROR:  Could not open file  %s", path); return NULL; }
    fseek(fp, 0, SEEK_END);
    sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    buf = (unsigned char *)malloc(sz > 0 ? (size_t)sz : 1);
    if (!buf || fread(buf, 1, (size_t)sz, fp) != (size_t)sz) {
        if (err) snprintf(err, errlen, "ERROR reading ICM header");
        free(buf); fclose(fp); return NULL;
    }
    fclose(fp);
    m = orc_model_from_bytes(buf, (size_t)sz, err, errlen);
    free(buf);
    return m;
}

/* icm.cc:729-753 (Output), :757-773 (Output_Node, binary), :961-998 (Write_Header):
 * the root of every sub-model is always written, other nodes only when
 * mip >= -1; int32 -1 closes the file. */
int orc_model_write(const orc_model *m, const char *path)
{
    FILE *fp = fopen(path, "wb");
    char line[ORC_ID_STRING_LEN];
    int32_t par[6], endmark = -1;
    int f, i;
    if (!fp) return -1;
    memset(line, 0, sizeof line);
    snprintf(line, sizeof line, ">ver = %.2f  len = %d  depth = %d  periodicity = %d  nodes = %d\n",
             ORC_VERSION_ID / 100.0, m->model_len, m->model_depth, m->periodicity, m->num_nodes);
    fwrite(line, 1, ORC_ID_STRING_LEN, fp);
    par[0] = ORC_VERSION_ID; par[1] = ORC_ID_STRING_LEN; par[2] = m->model_len;
    par[3] = m->model_depth; par[4] = m->periodicity; par[5] = m->num_nodes;
    fwrite(par, 4, 6, fp);
    for (f = 0; f < m->periodicity; f++)
        for (i = 0; i < m->num_nodes; i++) {
            size_t slot = (size_t)f * m->num_nodes + i;
            int32_t id = i;
            if (i > 0 && m->mip[slot] < -1) continue;
            fwrite(&id, 4, 1, fp);
            fwrite(m->prob + 4 * slot, 4, 4, fp);
            fwrite(m->mip + slot, 2, 1, fp);
        }
    fwrite(&endmark, 4, 1, fp);
    return fclose(fp) == 0 ? 0 : -1;
}

/* icm.cc:65-216.  64 codon probabilities from GC%, stop codons (spelled
 * backwards, because ORFs are scored 3'->5') forced to 1e-20, renormalised,
 * then marginalised into a 3 x 21-node tree.  The accumulators are the float
 * prob fields themselves, exactly as in the reference (float += double). */
int orc_build_indep_wo_stops(orc_model *m, double gc_frac, const char *const *stop_codon, int n_stops)
{
    double codon_prob[64], base_prob[4], sum;
    int i, j, k;
    static const int pw[3] = {1, 4, 16};

    if (m->model_len != 3 || m->model_depth != 2 || m->periodicity != 3 || m->num_nodes != 21)
        return -1;

    base_prob[1] = base_prob[2] = gc_frac / 2.0;
    base_prob[0] = base_prob[3] = 0.5 - base_prob[1];
    for (i = 0; i < 64; i++)   /* index = 16*first + 4*second + third (icm.cc:99-114) */
        codon_prob[i] = base_prob[(i >> 4) & 3] * base_prob[(i >> 2) & 3] * base_prob[i & 3];

    for (i = 0; i < n_stops; i++) {
        j = orc_subscript(stop_codon[i][0]) + 4 * orc_subscript(stop_codon[i][1])
            + 16 * orc_subscript(stop_codon[i][2]);
        codon_prob[j] = 1e-20;
    }
    sum = 0.0;
    for (i = 0; i < 64; i++) sum += codon_prob[i];
    for (i = 0; i < 64; i++) codon_prob[i] /= sum;

    memset(m->prob, 0, sizeof(float) * 4 * 3 * 21);
    /* NOTE: the reference does not reset mut_info_pos here; it relies on the
     * constructor's calloc (icm.cc:36-42).  Mirror that for a fresh model. */

    for (i = 0; i < 3; i++) {                       /* roots, icm.cc:149-162 */
        float *root = m->prob + 4 * (size_t)(i * 21);
        int d1 = pw[(3 - i) % 3];
        m->mip[i * 21] = (i == 1) ? -1 : 1;
        for (j = 0; j < 64; j++) root[(j / d1) % 4] += codon_prob[j];
    }
    for (i = 0; i < 3; i++) {                       /* level 1, icm.cc:165-180 */
        int d1 = pw[(3 - i) % 3], d2 = pw[(4 - i) % 3];
        for (j = 0; j < 4; j++) m->mip[i * 21 + 1 + j] = (i == 2) ? -1 : 0;
        if (i != 1)
            for (j = 0; j < 64; j++)
                m->prob[4 * (size_t)(i * 21 + 1 + (j / d2) % 4) + (j / d1) % 4] += codon_prob[j];
    }
    {                                               /* level 2, sub-model 0 only, icm.cc:185-199 */
        int d1 = pw[0], d2 = pw[1], d3 = pw[2];
        for (j = 0; j < 16; j++) m->mip[5 + j] = -1;
        for (j = 0; j < 64; j++) {
            k = 4 * ((j / d2) % 4) + (j / d3) % 4;
            m->prob[4 * (size_t)(5 + k) + (j / d1) % 4] += codon_prob[j];
        }
    }
    for (i = 0; i < 3; i++)                         /* normalise + log, icm.cc:202-211 */
        for (j = 0; j < 21; j++) {
            float *p = m->prob + 4 * (size_t)(i * 21 + j);
            sum = 0.0;
            for (k = 0; k < 4; k++) sum += p[k];
            for (k = 0; k < 4; k++) p[k] = (sum == 0.0 ? 0.0 : log(p[k] / sum));
        }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* per-base scoring                                                          */
/* ------------------------------------------------------------------------ */

/* shared descent of icm.cc:521-548 / 568-595; returns the node whose row is used */
static int orc_full_descent(const orc_model *m, const char *s, int frame)
{
    const int16_t *mip = m->mip + (size_t)frame * m->num_nodes;
    int node = 0, i, pos;
    for (i = 0; i < m->model_depth; i++) {
        pos = mip[node];
        if (pos == -1) break;
        if (pos < -1) { node = orc_parent(node); break; }
        node = node * ORC_ALPHA + orc_subscript(s[pos]) + 1;
    }
    if (mip[node] < -1) node = orc_parent(node);
    return node;
}

double orc_full_window_prob(const orc_model *m, const char *s, int frame)
{
    int node = orc_full_descent(m, s, frame);
    int sub = orc_subscript(s[m->model_len - 1]);
    return (double)m->prob[4 * ((size_t)frame * m->num_nodes + node) + sub];
}

void orc_full_window_distrib(const orc_model *m, const char *s, int frame, float dist[4])
{
    int node = orc_full_descent(m, s, frame);
    memcpy(dist, m->prob + 4 * ((size_t)frame * m->num_nodes + node), 4 * sizeof(float));
}

/* icm.cc:807-842: the window would start before the buffer; descend only while
 * the context position named by the node is inside the buffer. */
double orc_partial_window_prob(const orc_model *m, int predict_pos, const char *s, int frame)
{
    const int16_t *mip = m->mip + (size_t)frame * m->num_nodes;
    int start = predict_pos - (m->model_len - 1);
    int node = 0, i, pos;
    for (i = 0; i < m->model_depth; i++) {
        pos = start + mip[node];
        if (pos < 0) break;
        node = node * ORC_ALPHA + orc_subscript(s[pos]) + 1;
    }
    if (mip[node] == -2) node = orc_parent(node);
    return (double)m->prob[4 * ((size_t)frame * m->num_nodes + node) + orc_subscript(s[predict_pos])];
}

/* ------------------------------------------------------------------------ */
/* accumulations                                                             */
/* ------------------------------------------------------------------------ */

static int orc_next_frame(const orc_model *m, int frame)
{
    return (frame == m->periodicity - 1) ? 0 : frame + 1;
}

/* icm.cc:864-903 */
double orc_score_string(const orc_model *m, const char *s, int len, int frame)
{
    double result = 0.0;
    int i, start, stop = m->model_len - 1;
    if (m->periodicity == 1) frame = 0;
    for (i = 0; i < len && i < m->model_len - 1; i++) {
        result += orc_partial_window_prob(m, i, s, frame);
        frame = (frame + 1) % m->periodicity;
    }
    for (start = 0; stop < len; start++, stop++) {
        result += orc_full_window_prob(m, s + start, frame);
        frame = (frame + 1) % m->periodicity;
    }
    return result;
}

/* icm.cc:354-405 */
void orc_cumulative_score(const orc_model *m, const char *s, int n, double *score, int frame)
{
    double result = 0.0;
    int i, start, stop;
    if (m->periodicity == 1) frame = 0;
    stop = (m->model_len - 1 < n) ? m->model_len - 1 : n;
    for (i = 0; i < stop; i++) {
        result += orc_partial_window_prob(m, i, s, frame);
        frame = orc_next_frame(m, frame);
        score[i] = result;
    }
    for (start = 0; i < n; start++, i++) {
        result += orc_full_window_prob(m, s + start, frame);
        frame = orc_next_frame(m, frame);
        score[i] = result;
    }
}

/* icm.cc:409-452: like the above but shifted by one, cum_score[0] = 0, and the
 * partial loop always runs model_len-1 times (caller guarantees len >= that). */
void orc_cumulative_score_string(const orc_model *m, const char *s, int len, int frame, double *cum_score)
{
    double result;
    int i, start, stop = m->model_len - 1;
    if (m->periodicity == 1) frame = 0;
    result = cum_score[0] = 0.0;
    for (i = 0; i < m->model_len - 1; i++) {
        result += orc_partial_window_prob(m, i, s, frame);
        frame = (frame + 1) % m->periodicity;
        cum_score[i + 1] = result;
    }
    for (start = 0; stop < len; start++, stop++) {
        result += orc_full_window_prob(m, s + start, frame);
        frame = (frame + 1) % m->periodicity;
        cum_score[stop + 1] = result;
    }
}

/* icm.cc:485-509: one fixed sub-model for every position, no sum */
void orc_frame_score(const orc_model *m, const char *s, int n, double *score, int frame)
{
    int i, start, stop = (m->model_len - 1 < n) ? m->model_len - 1 : n;
    for (i = 0; i < stop; i++) score[i] = orc_partial_window_prob(m, i, s, frame);
    for (start = 0; i < n; start++, i++) score[i] = orc_full_window_prob(m, s + start, frame);
}

/* ------------------------------------------------------------------------ */
/* buffers and six-frame loops                                               */
/* ------------------------------------------------------------------------ */

/* glimmer_base.cc:2505-2533: s[start], s[start-1], ... (wraps below 0) */
void orc_reverse_transfer(char *buff, const char *s, int n, int start, int len)
{
    int j;
    for (j = 0; j < len; j++, start--) {
        buff[j] = s[start];
        if (start <= 0) start += n;
    }
    buff[len] = '\0';
}

/* glimmer_base.cc:410-434: complement of s[start], s[start+1], ... (wraps at n) */
void orc_complement_transfer(char *buff, const char *s, int n, int start, int len)
{
    int j;
    for (j = 0; j < len; j++, start++) {
        if (start >= n) start -= n;
        buff[j] = (char)orc_complement(s[start]);
    }
    buff[len] = '\0';
}

/* glimmer-mg.cc:1468-1510.  Rows 0-2: reversed read, sub-model f, written back
 * in forward coordinates; rows 3-5: complemented read. */
void orc_score_all_frames(const orc_model *gene, const orc_model *indep, const char *seq, int L, double *out)
{
    char *buff = (char *)malloc((size_t)L + 1);
    double *g = (double *)malloc(sizeof(double) * (size_t)(L ? L : 1));
    double *z = (double *)malloc(sizeof(double) * (size_t)(L ? L : 1));
    int f, i;
    if (L > 0) {
        orc_reverse_transfer(buff, seq, L, L - 1, L);
        for (f = 0; f < 3; f++) {
            orc_frame_score(gene, buff, L, g, f);
            orc_frame_score(indep, buff, L, z, f);
            for (i = 0; i < L; i++) out[(size_t)f * L + i] = g[L - 1 - i] - z[L - 1 - i];
        }
        orc_complement_transfer(buff, seq, L, 0, L);
        for (f = 0; f < 3; f++) {
            orc_frame_score(gene, buff, L, g, f);
            orc_frame_score(indep, buff, L, z, f);
            for (i = 0; i < L; i++) out[(size_t)(3 + f) * L + i] = g[i] - z[i];
        }
    }
    free(buff); free(g); free(z);
}

/* glimmer-mg.cc:561-604: forward ORF (frame > 0) walks from hi-1 downwards in
 * rows 1,2,0,...; reverse ORF walks from lo-1 upwards in rows 3+1,3+2,3+0,... */
void orc_cumulative_frame_score(const double *fs, int L, int frame, int lo, int hi, double *score)
{
    double cum = 0;
    int f = 1, len = hi - lo, i, si;
    if (frame > 0) {
        si = hi - 1;
        for (i = 0; i < len; i++) {
            score[i] = cum + fs[(size_t)f * L + si];
            cum = score[i];
            si--;
            f = (f == 2) ? 0 : f + 1;
        }
    } else {
        si = lo - 1;
        for (i = 0; i < len; i++) {
            score[i] = cum + fs[(size_t)(3 + f) * L + si];
            cum = score[i];
            si++;
            f = (f == 2) ? 0 : f + 1;
        }
    }
}

/* glimmer3.cc:328-359 with Permute_By_Frame (glimmer3.cc:1013-1088) written as
 * index tables: af_out[i] = af_raw[perm[i]]. */
void orc_all_frame_score(const orc_model *gene, const char *s, int len, int frame, double af[6])
{
    static const int perm_p1[6] = {2, 0, 1, 5, 3, 4};
    static const int perm_p2[6] = {1, 2, 0, 4, 5, 3};
    static const int perm_p3[6] = {0, 1, 2, 3, 4, 5};
    static const int perm_m1[6] = {3, 5, 4, 0, 2, 1};
    static const int perm_m2[6] = {4, 3, 5, 1, 0, 2};
    static const int perm_m3[6] = {5, 4, 3, 2, 1, 0};
    const int *perm = perm_p3;
    double raw[6];
    char *rc = (char *)malloc((size_t)len + 1);
    int i, j;

    raw[0] = orc_score_string(gene, s, len, 1);
    raw[1] = orc_score_string(gene, s, len, 2);
    raw[2] = orc_score_string(gene, s, len, 0);
    /* glimmer_base.cc:2484-2501 Reverse_Complement_Transfer(rc, s, 0, len) */
    for (j = 0, i = len - 1; i >= 0; j++, i--) rc[j] = (char)orc_complement(s[i]);
    rc[len] = '\0';
    raw[3] = orc_score_string(gene, rc, len, 1);
    raw[4] = orc_score_string(gene, rc, len, 0);
    raw[5] = orc_score_string(gene, rc, len, 2);
    free(rc);

    switch (frame) {
    case 1: perm = perm_p1; break;
    case 2: perm = perm_p2; break;
    case 3: perm = perm_p3; break;
    case -1: perm = perm_m1; break;
    case -2: perm = perm_m2; break;
    case -3: perm = perm_m3; break;
    }
    for (i = 0; i < 6; i++) af[i] = raw[perm[i]];
}

/* ------------------------------------------------------------------------ */
/* Score_Orfs inner loop                                                     */
/* ------------------------------------------------------------------------ */

/* gene.cc:954-995: one bit per base a possible IUPAC letter can be */
unsigned orc_ch_mask(int ch)
{
    switch (tolower(ch)) {
    case 'a': return 0x1; case 'c': return 0x2; case 'g': return 0x4; case 't': return 0x8;
    case 'r': return 0x5; case 'y': return 0xA; case 's': return 0x6; case 'w': return 0x9;
    case 'm': return 0x3; case 'k': return 0xC; case 'b': return 0xE; case 'd': return 0xD;
    case 'h': return 0xB; case 'v': return 0x7; case 'n': return 0xF;
    }
    return 0x0;
}

/* Codon_t::Set_From (gene.cc:133-146): three Shift_In's of a 12-bit register */
static unsigned orc_codon_from(const char *s)
{
    unsigned d = 0;
    int i;
    for (i = 0; i < 3 && s[i]; i++) d = ((d & 0xff) << 4) | orc_ch_mask(s[i]);
    return d;
}

/* Codon_t::Can_Be (gene.cc:39-66): every base position shares at least one possible letter */
static int orc_can_be(unsigned data, const unsigned *pat, int n, int *which)
{
    int i;
    for (i = 0; i < n; i++) {
        unsigned x = data & pat[i];
        if ((x & 0xf00) && (x & 0xf0) && (x & 0x0f)) { *which = i; return 1; }
    }
    *which = -1;
    return 0;
}

int orc_score_orf(const orc_model *gene, const orc_model *indep, const char *seq, int seq_len,
                  int frame, int stop_position, int orf_len, const orc_orf_params *prm,
                  orc_start *starts, int cap, orc_orf_out *out)
{
    unsigned pat[8], codon = 0;
    int n_pat = prm->n_start_codons, n_starts = 0;
    int len = orf_len, lo, hi, k, j, m, lowest_j, which = -1, i;
    int first_pos = 0, best_pos = 0, first_j = 0, best_j = 0;
    int orf_is_truncated, first_is_truncated = 0, best_is_truncated = 0;
    double first_score = -1.7976931348623157e308, best_score = -1.7976931348623157e308;   /* -DBL_MAX */
    char *buff = (char *)malloc((size_t)len + 1);
    double *score = (double *)malloc(sizeof(double) * (size_t)(len ? len : 1));
    double *indep_score = (double *)malloc(sizeof(double) * (size_t)(len ? len : 1));

    for (i = 0; i < n_pat; i++) pat[i] = orc_codon_from(prm->start_codon[i]);

    if (frame > 0) {                                    /* glimmer3.cc:1322-1332 */
        hi = stop_position - 1;
        lo = hi - len;
        orc_reverse_transfer(buff, seq, seq_len, hi - 1, len);
        orf_is_truncated = (lo < 3 && prm->allow_truncated);
        k = stop_position - len - 2;
    } else {                                            /* glimmer3.cc:1333-1343 */
        lo = stop_position + 2;
        hi = lo + len;
        orc_complement_transfer(buff, seq, seq_len, lo, len);
        orf_is_truncated = (seq_len - hi < 3 && prm->allow_truncated);
        k = stop_position + len + 4;
    }
    orc_cumulative_score(gene, buff, len, score, 1);    /* glimmer3.cc:1346-1347 */
    orc_cumulative_score(indep, buff, len, indep_score, 1);
    m = len;

    lowest_j = prm->min_gene_len - 3 < 3 ? prm->min_gene_len - 3 : 3;     /* Min (3, Min_Gene_Len - 3) */
    for (j = m - 1; j >= lowest_j; j--) {               /* glimmer3.cc:1355-1421 */
        codon = ((codon & 0xff) << 4) | orc_ch_mask(buff[j]);
        if (j % 3 == 0 && (orc_can_be(codon, pat, n_pat, &which) || (first_pos == 0 && orf_is_truncated))
            && j + 3 >= prm->min_gene_len) {
            double next_s = score[j - 1] - indep_score[j - 1];
            orc_start st;
            st.j = j + 2; st.pos = k; st.score = next_s; st.first = (first_pos == 0);
            if (which >= 0 && first_pos == 0 && orf_is_truncated) {
                st.which = -1; st.truncated = 1;
                if (n_starts < cap) starts[n_starts] = st;
                n_starts++;
                st.first = 0;
            }
            st.which = which; st.truncated = (which < 0);
            if (n_starts < cap) starts[n_starts] = st;
            n_starts++;
            if (first_pos == 0) {
                first_score = next_s; first_pos = k; first_j = j + 2;
                first_is_truncated = (first_pos == 0 && orf_is_truncated);
            }
            if (next_s > best_score) {
                best_score = next_s; best_pos = k; best_j = j + 2; best_is_truncated = st.truncated;
            }
        }
        if (frame > 0) k++; else k--;
    }
    if (prm->use_first_start) {                         /* glimmer3.cc:1423-1429 */
        best_score = first_score; best_pos = first_pos; best_j = first_j; best_is_truncated = first_is_truncated;
    }
    (void)best_is_truncated;
    free(buff); free(score); free(indep_score);
    out->first_j = first_j; out->best_j = best_j; out->best_pos = best_pos;
    out->best_score = best_score; out->orf_is_truncated = orf_is_truncated;
    out->is_tentative_gene = 0; out->gene_score = 0.0;
    if (first_j + 1 < prm->min_gene_len) return -1;     /* glimmer3.cc:1431-1432 */
    for (i = 0; i < n_starts && i < cap; i++)           /* glimmer3.cc:1464-1466 */
        if (starts[i].j > prm->ignore_score_len && starts[i].score < 0.0) starts[i].score = 0.0;
    out->is_tentative_gene = (first_j + 1 >= prm->min_gene_len && best_score > prm->start_threshold);
    out->gene_score = 100.0 * best_score / (best_j - 2);
    return n_starts;
}

/* ======================================================================================================
 * glimmer-mg front half: Find_Orfs, Save_Prev_Stops, Score_Orf_Starts (no errors), Score_Orfs_Errors filter
 * ==================================================================================================== */

/* Codon_t::Must_Be (gene.cc:70-92): every string this codon could be matches the pattern */
static int orc_must_be(unsigned data, const unsigned *pat, int n)
{
    int i;
    for (i = 0; i < n; i++)
        if ((data & pat[i]) == data && (data & 0xf00) && (data & 0xf0) && (data & 0x0f)) return 1;
    return 0;
}

/* Codon_t::Reverse_Complement (gene.cc:96-113): the 12 bits mirrored */
static unsigned orc_codon_revcomp(unsigned data)
{
    unsigned x = 0;
    int i;
    for (i = 0; i < 12; i++) { x = (x << 1) | (data & 1u); data >>= 1; }
    return x;
}

#define ORC_INT_MAX 2147483647

typedef struct orc_find_state {
    const orc_mg_params *prm;
    orc_orf *orfs;
    int cap, n_orfs;
    int min_indel_orf_len;                              /* < 0: Allow_Indels = Allow_Subs = false */
    int first_fwd_start[3], last_rev_start[3], prev_fwd_stop[3], prev_rev_stop[3];
} orc_find_state;

static void orc_push_orf(orc_find_state *st, int stop_position, int frame, int gene_len, int orf_len)
{
    /* glimmer_base.cc:494,528,806 */
    if (gene_len >= st->prm->min_gene_len || (st->min_indel_orf_len >= 0 && orf_len >= st->min_indel_orf_len)) {
        if (st->n_orfs < st->cap) {
            orc_orf *o = &st->orfs[st->n_orfs];
            o->stop_position = stop_position; o->frame = frame; o->gene_len = gene_len; o->orf_len = orf_len;
        }
        st->n_orfs++;
    }
}

/* Do_Fwd_Stop_Codon (glimmer_base.cc:460-504) + Handle_First_Forward_Stop without wrap-around (:946-985) */
static void orc_do_fwd_stop(orc_find_state *st, int i, int frame)
{
    int gene_len, orf_len;
    if (st->prev_fwd_stop[frame] == 0) {
        const int pos = i - 1, start_pos = st->first_fwd_start[frame], first_base = 1;
        orf_len = pos - first_base;
        orf_len -= orf_len % 3;
        gene_len = (start_pos == ORC_INT_MAX) ? 0 : pos - start_pos;
        if (st->prm->allow_truncated && gene_len < st->prm->min_gene_len) gene_len = orf_len;
    } else {
        gene_len = i - st->first_fwd_start[frame] - 1;
        orf_len = i - st->prev_fwd_stop[frame] - 4;
    }
    orc_push_orf(st, i - 1, 1 + (frame + 1) % 3, gene_len, orf_len);
    st->first_fwd_start[frame] = ORC_INT_MAX;
    st->prev_fwd_stop[frame] = i - 1;
}

/* Do_Rev_Stop_Codon (glimmer_base.cc:506-537) + Handle_First_Reverse_Stop (:989-1015) */
static void orc_do_rev_stop(orc_find_state *st, int i, int frame)
{
    int gene_len, orf_len, orf_stop = 0;
    if (st->prev_rev_stop[frame] == 0) {
        if (!st->prm->allow_truncated) gene_len = 0;
        else {
            orf_stop = (i - 1) % 3;
            if (orf_stop > 0) orf_stop -= 3;
            gene_len = st->last_rev_start[frame] - orf_stop;
        }
    } else {
        orf_stop = st->prev_rev_stop[frame];
        gene_len = st->last_rev_start[frame] - orf_stop;
    }
    orf_len = i - orf_stop - 4;
    orc_push_orf(st, orf_stop, -1 - (frame + 1) % 3, gene_len, orf_len);
    st->last_rev_start[frame] = 0;
    st->prev_rev_stop[frame] = i - 1;
}

int orc_find_orfs(const char *seq, int n, const orc_mg_params *prm, orc_orf *orfs, int cap)
{
    return orc_find_orfs_err(seq, n, prm, -1, orfs, cap);
}

int orc_find_orfs_err(const char *seq, int n, const orc_mg_params *prm, int min_indel_orf_len, orc_orf *orfs, int cap)
{
    orc_find_state st;
    unsigned fwd_start[8], rev_start[8], fwd_stop[8], rev_stop[8], codon = 0;
    int i, frame, fr;
    st.prm = prm; st.orfs = orfs; st.cap = cap; st.n_orfs = 0; st.min_indel_orf_len = min_indel_orf_len;
    for (i = 0; i < 3; i++) {
        st.first_fwd_start[i] = ORC_INT_MAX;
        st.last_rev_start[i] = st.prev_fwd_stop[i] = st.prev_rev_stop[i] = 0;
    }
    for (i = 0; i < prm->n_start_codons; i++) {       /* Set_Start_And_Stop_Codons, glimmer_base.cc:2688-2704 */
        fwd_start[i] = orc_codon_from(prm->start_codon[i]);
        rev_start[i] = orc_codon_revcomp(fwd_start[i]);
    }
    for (i = 0; i < prm->n_stop_codons; i++) {
        fwd_stop[i] = orc_codon_from(prm->stop_codon[i]);
        rev_stop[i] = orc_codon_revcomp(fwd_stop[i]);
    }
    if (n < prm->min_gene_len) return 0;                /* glimmer_base.cc:676-677 */

    frame = 0;
    for (i = 0; i < n; i++) {                           /* glimmer_base.cc:701-757, no ignore regions */
        int which;
        codon = ((codon & 0xff) << 4) | orc_ch_mask(seq[i]);
        if (orc_can_be(codon, fwd_start, prm->n_start_codons, &which) && st.first_fwd_start[frame] == ORC_INT_MAX)
            st.first_fwd_start[frame] = i - 1;
        if (orc_can_be(codon, rev_start, prm->n_start_codons, &which)) st.last_rev_start[frame] = i - 1;
        if (orc_must_be(codon, fwd_stop, prm->n_stop_codons)) orc_do_fwd_stop(&st, i, frame);
        if (orc_must_be(codon, rev_stop, prm->n_stop_codons)) orc_do_rev_stop(&st, i, frame);
        frame = frame == 2 ? 0 : frame + 1;
    }
    /* Finish_Orfs (glimmer_base.cc:783-817) + Handle_Last_Reverse_Stop without wrap-around (:1019-1072) */
    for (fr = 0; fr < 3; fr++) {
        int orf_stop, orf_len, gene_len;
        if (st.prev_rev_stop[fr] == 0) orf_stop = fr == 0 ? -1 : fr == 1 ? 0 : -2;
        else orf_stop = st.prev_rev_stop[fr];
        orf_len = n - orf_stop - 2;
        orf_len -= orf_len % 3;
        gene_len = st.last_rev_start[fr] == 0 ? 0 : st.last_rev_start[fr] - orf_stop;
        if (prm->allow_truncated && gene_len < prm->min_gene_len) gene_len = orf_len;
        orc_push_orf(&st, orf_stop, -1 - (fr + 1) % 3, gene_len, orf_len);
    }
    if (prm->allow_truncated)                           /* glimmer_base.cc:765-776: 3 bp past the end are stops */
        for (; i < n + 3; i++) {
            orc_do_fwd_stop(&st, i, frame);
            frame = frame == 2 ? 0 : frame + 1;
        }
    return st.n_orfs;
}

/* ---- Find_Orfs in full: ignore regions (glimmer3 -i) and circular sequences (glimmer-mg -r) ------------------------
 * glimmer_base.cc:638-779 with Do_Fwd_Stop_Codon :460-504, Do_Rev_Stop_Codon :506-537, Finish_Orfs :783-817,
 * Handle_First_Forward_Stop :946-985, Handle_First_Reverse_Stop :989-1015, Handle_Last_Reverse_Stop :1019-1072,
 * Wrap_Around_Back :2793-2850, Wrap_Through_Front :2854-2900.  ign_lo / ign_hi: the regions as Get_Ignore_Regions
 * (:833-930) leaves them (0-based lo, hi = one past the last ignored base; sorted, overlaps merged).
 * Returns the number of ORFs, or -1 where the reference's assert (pos > 0) in Wrap_Around_Back would fire. */
typedef struct orc_fg {
    orc_find_state st;
    const char *seq;
    int len;                                            /* Sequence_Len */
    int circular, hit_ignore, first_base, failed;
    const unsigned *fwd_start, *rev_start, *fwd_stop, *rev_stop;
} orc_fg;

/* Codon_t::Reverse_Shift_In (gene.cc): the new base becomes the codon's FIRST position */
static unsigned orc_rev_shift(unsigned codon, int ch) { return (codon >> 4) | (orc_ch_mask(ch) << 8); }

static void orc_wrap_through_front(orc_fg *g, int pos, int *gene_len, int *orf_len)
{
    unsigned codon = 0;
    int start_at = -1, s = (pos - 1) % 3, check_len = g->len + s - pos - 4, i, j, which;
    for (i = 0; i < check_len; i += 3) {
        for (j = 0; j < 3; j++) {
            s--;
            if (s < 0) s += g->len;
            codon = orc_rev_shift(codon, g->seq[s]);
        }
        if (orc_must_be(codon, g->fwd_stop, g->st.prm->n_stop_codons)) break;
        if (orc_can_be(codon, g->fwd_start, g->st.prm->n_start_codons, &which)) start_at = i + 3;
    }
    *orf_len = i + 3 * ((pos - 1) / 3);
    *gene_len = start_at == -1 ? 0 : start_at + 3 * ((pos - 1) / 3);
}

static void orc_wrap_around_back(orc_fg *g, int wfr, int pos, int *gene_len, int *orf_len)
{
    unsigned codon = 0;
    int start_at = -1, orf_add = 0, frame = 0, check_len = pos - 1, i, which;
    if (pos <= 0) { g->failed = 1; *gene_len = *orf_len = 0; return; }     /* the reference: assert (pos > 0) */
    for (i = 0; i < check_len; i++) {
        codon = ((codon & 0xff) << 4) | orc_ch_mask(g->seq[i]);
        if (frame == wfr) {
            if (orc_must_be(codon, g->rev_stop, g->st.prm->n_stop_codons)) { orf_add = i - 2; break; }
            orf_add = i + 1;
        }
        if (frame == wfr && orc_can_be(codon, g->rev_start, g->st.prm->n_start_codons, &which)) start_at = i + 1;
        frame = frame == 2 ? 0 : frame + 1;
    }
    *orf_len = orf_add + g->len - pos - 2;
    *orf_len -= *orf_len % 3;
    *gene_len = start_at == -1 ? 0 : start_at + g->len - pos - 2;
}

static void orc_fg_fwd_stop(orc_fg *g, int i, int frame)
{
    orc_find_state *st = &g->st;
    int gene_len, orf_len;
    if (st->prev_fwd_stop[frame] == 0) {
        const int pos = i - 1, start_pos = st->first_fwd_start[frame];
        if (g->circular && !g->hit_ignore) {            /* Handle_First_Forward_Stop with use_wraparound */
            orc_wrap_through_front(g, pos, &gene_len, &orf_len);
            if (gene_len == 0 && start_pos != ORC_INT_MAX) gene_len = pos - start_pos;
        } else {
            orf_len = pos - g->first_base;
            orf_len -= orf_len % 3;
            gene_len = start_pos == ORC_INT_MAX ? 0 : pos - start_pos;
            if (st->prm->allow_truncated && gene_len < st->prm->min_gene_len) gene_len = orf_len;
        }
    } else {
        gene_len = i - st->first_fwd_start[frame] - 1;
        orf_len = i - st->prev_fwd_stop[frame] - 4;
    }
    orc_push_orf(st, i - 1, 1 + (frame + 1) % 3, gene_len, orf_len);
    st->first_fwd_start[frame] = ORC_INT_MAX;
    st->prev_fwd_stop[frame] = i - 1;
}

static void orc_fg_rev_stop(orc_fg *g, int i, int frame)
{
    orc_find_state *st = &g->st;
    int gene_len, orf_len, orf_stop = 0;
    if (st->prev_rev_stop[frame] == 0) {
        if (g->hit_ignore || !st->prm->allow_truncated) gene_len = 0;
        else {
            orf_stop = (i - 1) % 3;
            if (orf_stop > 0) orf_stop -= 3;
            gene_len = st->last_rev_start[frame] - orf_stop;
        }
    } else {
        orf_stop = st->prev_rev_stop[frame];
        gene_len = st->last_rev_start[frame] - orf_stop;
    }
    orf_len = i - orf_stop - 4;
    orc_push_orf(st, orf_stop, -1 - (frame + 1) % 3, gene_len, orf_len);
    st->last_rev_start[frame] = 0;
    st->prev_rev_stop[frame] = i - 1;
}

static void orc_fg_finish(orc_fg *g, int use_wraparound, int last_position)
{
    orc_find_state *st = &g->st;
    int fr;
    for (fr = 0; fr < 3; fr++) {
        int orf_stop, orf_len, gene_len;
        if (st->prev_rev_stop[fr] == 0) orf_stop = fr == 0 ? -1 : fr == 1 ? 0 : -2;
        else orf_stop = st->prev_rev_stop[fr];
        if (use_wraparound) {
            const int wrap_fr = (3 + fr - (g->len % 3)) % 3;
            orc_wrap_around_back(g, wrap_fr, st->prev_rev_stop[fr], &gene_len, &orf_len);
            if (gene_len == 0 && st->last_rev_start[fr] > 0) gene_len = st->last_rev_start[fr] - st->prev_rev_stop[fr];
        } else {
            orf_len = last_position - orf_stop - 2;
            orf_len -= orf_len % 3;
            gene_len = st->last_rev_start[fr] == 0 ? 0 : st->last_rev_start[fr] - orf_stop;
            if (st->prm->allow_truncated && gene_len < st->prm->min_gene_len) gene_len = orf_len;
        }
        orc_push_orf(st, orf_stop, -1 - (fr + 1) % 3, gene_len, orf_len);
    }
}

int orc_find_orfs_general(const char *seq, int len, const orc_mg_params *prm, int min_indel_orf_len, int circular,
                          const int *ign_lo, const int *ign_hi, int n_ignore, orc_orf *orfs, int cap)
{
    orc_fg g;
    unsigned fwd_start[8], rev_start[8], fwd_stop[8], rev_stop[8], codon = 0;
    int i, j, n = len, frame = 0, ignoring = 0, ignore_sub = 0, ignore_start, ignore_stop, which;
    g.st.prm = prm; g.st.orfs = orfs; g.st.cap = cap; g.st.n_orfs = 0; g.st.min_indel_orf_len = min_indel_orf_len;
    g.seq = seq; g.len = len; g.circular = circular; g.hit_ignore = 0; g.first_base = 1; g.failed = 0;
    g.fwd_start = fwd_start; g.rev_start = rev_start; g.fwd_stop = fwd_stop; g.rev_stop = rev_stop;
    for (i = 0; i < 3; i++) {
        g.st.first_fwd_start[i] = ORC_INT_MAX;
        g.st.last_rev_start[i] = g.st.prev_fwd_stop[i] = g.st.prev_rev_stop[i] = 0;
    }
    for (i = 0; i < prm->n_start_codons; i++) { fwd_start[i] = orc_codon_from(prm->start_codon[i]); rev_start[i] = orc_codon_revcomp(fwd_start[i]); }
    for (i = 0; i < prm->n_stop_codons; i++) { fwd_stop[i] = orc_codon_from(prm->stop_codon[i]); rev_stop[i] = orc_codon_revcomp(fwd_stop[i]); }
    if (len < prm->min_gene_len) return 0;
    if (circular) n += 2;                               /* two bases of overhang: codons that span the end */
    ignore_start = ignore_stop = ORC_INT_MAX;
    if (n_ignore > 0) { ignore_start = ign_lo[0]; ignore_stop = ign_hi[0]; }
    for (i = 0; i < n; i++) {
        if (i == ignore_start) {
            orc_fg_finish(&g, 0, i);
            g.hit_ignore = ignoring = 1;
        } else if (i == ignore_stop) {
            for (j = 0; j < 3; j++) {
                g.st.first_fwd_start[j] = ORC_INT_MAX;
                g.st.last_rev_start[j] = g.st.prev_fwd_stop[j] = g.st.prev_rev_stop[j] = 0;
            }
            codon = 0;
            g.first_base = i + 1;
            ignoring = 0;
            ignore_sub++;
            if (ignore_sub >= n_ignore) ignore_start = ignore_stop = ORC_INT_MAX;
            else { ignore_start = ign_lo[ignore_sub]; ignore_stop = ign_hi[ignore_sub]; }
        }
        if (!ignoring) {
            codon = ((codon & 0xff) << 4) | orc_ch_mask(seq[i < len ? i : i - len]);
            if (orc_can_be(codon, fwd_start, prm->n_start_codons, &which) && g.st.first_fwd_start[frame] == ORC_INT_MAX)
                g.st.first_fwd_start[frame] = i - 1;
            if (orc_can_be(codon, rev_start, prm->n_start_codons, &which)) g.st.last_rev_start[frame] = i - 1;
            if (orc_must_be(codon, fwd_stop, prm->n_stop_codons)) orc_fg_fwd_stop(&g, i, frame);
            if (orc_must_be(codon, rev_stop, prm->n_stop_codons)) orc_fg_rev_stop(&g, i, frame);
        }
        frame = frame == 2 ? 0 : frame + 1;
    }
    orc_fg_finish(&g, circular, len);
    if (!circular && prm->allow_truncated)
        for (; i < n + 3; i++) {
            if (!ignoring) orc_fg_fwd_stop(&g, i, frame);
            frame = frame == 2 ? 0 : frame + 1;
        }
    return g.failed ? -1 : g.st.n_orfs;
}

void orc_save_prev_stops(const char *seq, int n, const orc_mg_params *prm, int *fwd_prev, int *rev_next)
{
    unsigned fwd_stop[8], codon = 0;                    /* one Codon_t for both loops, like the reference */
    int last_stops[3] = {0, 1, -1}, frame = 0, i;
    for (i = 0; i < prm->n_stop_codons; i++) fwd_stop[i] = orc_codon_from(prm->stop_codon[i]);
    for (i = 0; i < n; i++) {                           /* glimmer-mg.cc:689-702 */
        codon = ((codon & 0xff) << 4) | orc_ch_mask(seq[i]);
        if (i >= 2 && orc_must_be(codon, fwd_stop, prm->n_stop_codons)) last_stops[frame] = i;
        fwd_prev[i] = last_stops[frame];
        frame = (frame + 1) % 3;
    }
    last_stops[0] = n - 1; last_stops[1] = n - 2; last_stops[2] = n;      /* glimmer-mg.cc:708-710 */
    frame = 0;
    for (i = n - 1; i >= 0; i--) {                      /* glimmer-mg.cc:714-728 */
        codon = ((codon & 0xff) << 4) | orc_ch_mask(orc_complement(seq[i]));
        if (i <= n - 3 && orc_must_be(codon, fwd_stop, prm->n_stop_codons)) last_stops[frame] = i;
        rev_next[i] = last_stops[frame];
        frame = (frame + 1) % 3;
    }
}

int orc_mg_score_orf(const double *frame_scores, const char *seq, int n, const int *fwd_prev, const int *rev_next,
                     int frame, int stop_position, const orc_mg_params *prm, orc_start *starts, int cap,
                     orc_mg_out *out)
{
    unsigned pat[8], codon = 0;
    int n_pat = prm->n_start_codons, n_starts = 0, i;
    int end_point, lo, hi, len, k, j, m, lowest_j, which = -1, first_pos = 0, orf_is_truncated;
    char *buff;
    double *score;
    for (i = 0; i < n_pat; i++) pat[i] = orc_codon_from(prm->start_codon[i]);

    if (frame > 0) {                                    /* glimmer-mg.cc:1637-1640, 1721-1742 */
        int e;
        end_point = stop_position - 1;
        hi = end_point;
        e = end_point - 1;                              /* Fwd_Prev_Stop, :642-652 */
        lo = ((e >= 0 && e < n) ? fwd_prev[e] : e) + 1;
        len = hi - lo;
        orf_is_truncated = (lo < 3 && prm->allow_truncated);
        k = lo - 1;
    } else {                                            /* glimmer-mg.cc:1744-1763 */
        int e;
        end_point = stop_position + 3;
        lo = end_point;
        e = end_point - 1;                              /* Rev_Next_Stop, :1436-1445 */
        hi = ((e >= 0 && e < n) ? rev_next[e] : e) + 1;
        len = hi - lo;
        orf_is_truncated = (n - (hi - 1) < 3 && prm->allow_truncated);
        k = hi + 1;
    }
    out->lo = lo; out->hi = hi; out->orf_is_truncated = orf_is_truncated;
    out->first_j = 0; out->accepted = 0; out->best_score = -1.7976931348623157e308;
    if (len < 0) return 0;
    buff = (char *)malloc((size_t)len + 1);
    score = (double *)malloc(sizeof(double) * (size_t)(len ? len : 1));
    if (frame > 0) orc_reverse_transfer(buff, seq, n, hi - 1, len);
    else orc_complement_transfer(buff, seq, n, lo - 1, len);
    orc_cumulative_frame_score(frame_scores, n, frame, lo, hi, score);     /* indep_score is all zero */

    m = len;
    lowest_j = prm->min_gene_len - 3 < 3 ? prm->min_gene_len - 3 : 3;
    for (j = m - 1; j >= lowest_j; j--) {               /* glimmer-mg.cc:1813-1860, suffix_j = 0, suffix_score = 0 */
        codon = ((codon & 0xff) << 4) | orc_ch_mask(buff[j]);
        if (j % 3 == 0 && (orc_can_be(codon, pat, n_pat, &which) || (first_pos == 0 && orf_is_truncated))
            && j + 3 >= prm->min_gene_len) {
            orc_start st;
            st.score = score[j - 1] - 0.0;
            st.j = j + 2; st.pos = k; st.first = (first_pos == 0);
            if (which >= 0 && first_pos == 0 && orf_is_truncated) {
                st.which = -1; st.truncated = 1;
                if (n_starts < cap) starts[n_starts] = st;
                n_starts++;
                st.first = 0;
            }
            st.which = which; st.truncated = (which < 0);
            if (n_starts < cap) starts[n_starts] = st;
            n_starts++;
            if (first_pos == 0) first_pos = k;
        }
        if (frame > 0) k++; else k--;
    }
    free(buff); free(score);

    /* Score_Orfs_Errors (glimmer-mg.cc:1647-1683) */
    for (i = 0; i < n_starts && i < cap; i++)           /* boost long ORFs: score = Max (0.0, score) */
        if (starts[i].j > prm->ignore_score_len && 0.0 > starts[i].score) starts[i].score = 0.0;
    if (n_starts > 0 && n_starts <= cap) {
        /* after sort by pos: forward -> front() has the lowest pos, reverse -> back() has the highest;
         * entries with equal pos (truncated + real start at the same codon) share their j */
        int pick = 0;
        double best = -1.7976931348623157e308;
        for (i = 1; i < n_starts; i++)
            if (frame > 0 ? starts[i].pos < starts[pick].pos : starts[i].pos > starts[pick].pos) pick = i;
        out->first_j = starts[pick].j;
        if (out->first_j + 1 >= prm->min_gene_len) {
            for (i = 0; i < n_starts; i++)
                if (starts[i].score > best) best = starts[i].score;
            out->best_score = best;
            out->accepted = best > prm->start_threshold;
        }
    }
    return n_starts;
}


/* ======================================================================================================
 * glimmer-mg's error branch: Score_Orf_Starts with indels (-i) or substitutions (-s), glimmer-mg.cc:1513-1861
 * ==================================================================================================== */

/* Set_Quality_454 (glimmer-mg.cc:1865-1906): the last base of a homopolymer run of length r gets
 * 31 - 5 r (r < 6) or 6; the bases inside a run 31 */
void orc_set_quality_454(const char *seq, int n, int *q)
{
    int run_q[6], i, run = 0;
    char last = ' ';
    for (i = 0; i < 6; i++) run_q[i] = 31 - 5 * i;
    if (n <= 0) return;
    for (i = 0; i < n; i++) {
        if (seq[i] != last) {
            if (i > 0) q[i - 1] = run < 6 ? run_q[run] : run_q[5];
            run = 1;
        } else {
            q[i - 1] = 31;
            run++;
        }
        last = seq[i];
    }
    q[n - 1] = run < 6 ? run_q[run] : run_q[5];
}

/* Clean_Quality_454 (glimmer-mg.cc:519-546) */
void orc_clean_quality_454(const char *seq, int n, int *q, int indel_quality_threshold)
{
    int i;
    for (i = 0; i < n; i++)
        if (q[i] <= 0) q[i] = 1;
    for (i = 1; i < n; i++)
        if (seq[i] == seq[i - 1] && q[i - 1] < indel_quality_threshold + 1) q[i - 1] = indel_quality_threshold + 1;
}

typedef struct orc_err_ctx {
    const double *fs; const char *seq; int n; const int *fwd_prev, *rev_next, *quality;
    int frame, stop_position;
    const orc_mg_params *prm; const orc_mg_err_params *ep;
    unsigned pat[8];
    orc_start_err *starts; int cap, n_starts;
    int lo0, hi0, trunc0, have0;
} orc_err_ctx;

/* Pass_Stop_Penalty (glimmer-mg.cc:961-995) without a quality file (with -s the reference never loads
 * Quality_Values, :384-392, so only default_p is defined behaviour) */
static double orc_pass_stop_penalty(const orc_err_ctx *c, int lo, int hi)
{
    const double default_p = 0.999;
    double codon_p[3], p_stop;
    int stop_i[3];
    codon_p[0] = codon_p[1] = codon_p[2] = default_p;
    stop_i[0] = lo - 3; stop_i[1] = lo - 2; stop_i[2] = lo - 1;
    if (c->frame < 0) { stop_i[0] = hi + 1; stop_i[1] = hi; stop_i[2] = hi - 1; }
    p_stop = codon_p[0];
    /* (an index outside the read reads outside the reference's string; callers keep it inside, see the test) */
    if ((c->frame > 0 && stop_i[1] >= 0 && stop_i[1] < c->n && c->seq[stop_i[1]] == 'a') ||
        (c->frame < 0 && stop_i[1] >= 0 && stop_i[1] < c->n && c->seq[stop_i[1]] == 't'))
        p_stop *= 2.0 / 3.0 * codon_p[1] + 1.0 / 3.0;
    else
        p_stop *= codon_p[1];
    if ((c->frame > 0 && stop_i[2] >= 0 && stop_i[2] < c->n && c->seq[stop_i[2]] == 'a') ||
        (c->frame < 0 && stop_i[2] >= 0 && stop_i[2] < c->n && c->seq[stop_i[2]] == 't'))
        p_stop *= 2.0 / 3.0 * codon_p[2] + 1.0 / 3.0;
    else
        p_stop *= codon_p[2];
    return log(1.0 - p_stop) - log(p_stop);
}

/* Score_Orf_Starts (glimmer-mg.cc:1693-1861), recursive exactly like the reference */
static void orc_score_orf_starts(orc_err_ctx *c, int end_point, double suffix_score, int suffix_j,
                                 const int *err_pos, const int *err_type, int num_errors)
{
    const orc_mg_params *prm = c->prm;
    const orc_mg_err_params *ep = c->ep;
    const int frame = c->frame, n = c->n;
    unsigned codon = 0;
    int lo, hi, len, k, j, m, lowest_j, which = -1, first_pos = 0, orf_is_truncated, e;
    char *buff;
    int *qbuf;
    double *score;
    int epos[ORC_MAX_ERRORS], etype[ORC_MAX_ERRORS];
    for (e = 0; e < num_errors; e++) { epos[e] = err_pos[e]; etype[e] = err_type[e]; }

    if (frame > 0) {                                    /* :1721-1742 */
        hi = end_point;
        e = end_point - 1;
        lo = ((e >= 0 && e < n) ? c->fwd_prev[e] : e) + 1;
        len = hi - lo;
        orf_is_truncated = (lo < 3 && prm->allow_truncated);
        k = lo - 1;
    } else {                                            /* :1744-1763 */
        lo = end_point;
        e = end_point - 1;
        hi = ((e >= 0 && e < n) ? c->rev_next[e] : e) + 1;
        len = hi - lo;
        orf_is_truncated = (n - (hi - 1) < 3 && prm->allow_truncated);
        k = hi + 1;
    }
    if (!c->have0) { c->lo0 = lo; c->hi0 = hi; c->trunc0 = orf_is_truncated; c->have0 = 1; }
    if (len < 0) len = 0;
    buff = (char *)malloc((size_t)len + 1);
    qbuf = (int *)malloc(sizeof(int) * (size_t)(len + 1));
    score = (double *)malloc(sizeof(double) * (size_t)(len + 1));
    if (frame > 0) {
        orc_reverse_transfer(buff, c->seq, n, hi - 1, len);
        for (j = 0; j < len; j++) qbuf[j] = c->quality ? c->quality[hi - 1 - j] : 0;       /* Reverse_Transfer_Qual */
    } else {
        if (lo - 1 < n) orc_complement_transfer(buff, c->seq, n, lo - 1, len);
        for (j = 0; j < len; j++) qbuf[j] = c->quality ? c->quality[lo - 1 + j] : 0;       /* Complement_Transfer_Qual */
    }
    orc_cumulative_frame_score(c->fs, n, frame, lo, hi, score);

    if (ep->allow_subs && num_errors < 1) {             /* mutate the previous stop codon, :1771-1806 */
        int error_end_point, error_pos;
        if (frame > 0) { error_end_point = lo - 3; error_pos = lo - 2; }
        else { error_end_point = hi + 3; error_pos = hi + 2; }
        if (error_end_point >= 0 && error_end_point - 2 < n) {
            const int error_suffix_j = suffix_j + len;
            double error_suffix_score = suffix_score + orc_pass_stop_penalty(c, lo, hi);
            if (len > 0) error_suffix_score += score[len - 1] - 0.0;
            epos[num_errors] = error_pos; etype[num_errors] = 2;
            orc_score_orf_starts(c, error_end_point, error_suffix_score, error_suffix_j, epos, etype, num_errors + 1);
        }
    }

    m = len;
    lowest_j = prm->min_gene_len - 3 < 3 ? prm->min_gene_len - 3 : 3;
    for (j = m - 1; j >= lowest_j; j--) {               /* :1813-1860 */
        if (ep->allow_indels && qbuf[j] <= ep->indel_quality_threshold && num_errors < ep->indel_max) {
            /* Score_Indels (:1513-1602) */
            const int q = qbuf[j];
            const double prob_err = pow(10.0, -(double)q / 10.0);
            const double score_penalty = log(prob_err / 2.0) - log(1.0 - prob_err);
            double es;
            if (frame > 0) {
                es = suffix_score + score[j] - 0.0 + score_penalty;
                if (es > ep->indel_suffix_score_threshold) {
                    epos[num_errors] = k + 3; etype[num_errors] = 1;
                    orc_score_orf_starts(c, k + (j % 3), es, suffix_j + j + 2 - (j % 3), epos, etype, num_errors + 1);
                }
                es = suffix_score + score[j - 1] - 0.0 + score_penalty;
                if (es > ep->indel_suffix_score_threshold) {
                    epos[num_errors] = k + 2; etype[num_errors] = 0;
                    orc_score_orf_starts(c, k - (2 - (j % 3)), es, suffix_j + j + 2 - (j % 3), epos, etype, num_errors + 1);
                }
            } else {
                es = suffix_score + score[j] - 0.0 + score_penalty;
                if (es > ep->indel_suffix_score_threshold) {
                    epos[num_errors] = k - 1; etype[num_errors] = 1;
                    orc_score_orf_starts(c, k - (j % 3), es, suffix_j + j + 2 - (j % 3), epos, etype, num_errors + 1);
                }
                es = suffix_score + score[j - 1] - 0.0 + score_penalty;
                if (es > ep->indel_suffix_score_threshold) {
                    epos[num_errors] = k - 2; etype[num_errors] = 0;
                    orc_score_orf_starts(c, k + 2 - (j % 3), es, suffix_j + j + 2 - (j % 3), epos, etype, num_errors + 1);
                }
            }
        }
        codon = ((codon & 0xff) << 4) | orc_ch_mask(buff[j]);
        if (j % 3 == 0 && (orc_can_be(codon, c->pat, prm->n_start_codons, &which) || (first_pos == 0 && orf_is_truncated))
            && j + 3 + suffix_j >= prm->min_gene_len) {
            orc_start_err st;
            const double next_s = score[j - 1] - 0.0;
            st.s.j = j + 2 + suffix_j; st.s.pos = k;
            st.s.score = next_s + suffix_score;
            st.s.first = (first_pos == 0);
            st.n_errors = num_errors;
            for (e = 0; e < ORC_MAX_ERRORS; e++) { st.err_pos[e] = e < num_errors ? epos[e] : 0; st.err_type[e] = e < num_errors ? etype[e] : 0; }
            if (which >= 0 && first_pos == 0 && orf_is_truncated) {
                st.s.which = -1; st.s.truncated = 1;
                if (c->n_starts < c->cap) c->starts[c->n_starts] = st;
                c->n_starts++;
                st.s.first = 0;
            }
            st.s.which = which; st.s.truncated = (which < 0);
            if (c->n_starts < c->cap) c->starts[c->n_starts] = st;
            c->n_starts++;
            if (first_pos == 0) first_pos = k;
        }
        if (frame > 0) k++; else k--;
    }
    free(buff); free(qbuf); free(score);
}

int orc_mg_score_orf_errors(const double *frame_scores, const char *seq, int n, const int *fwd_prev, const int *rev_next,
                            const int *quality, int frame, int stop_position, const orc_mg_params *prm,
                            const orc_mg_err_params *ep, orc_start_err *starts, int cap, orc_mg_out *out)
{
    orc_err_ctx c;
    int i, lo_pos, hi_pos, amb = 0, pick = -1;
    double best = -1.7976931348623157e308;
    memset(&c, 0, sizeof c);
    c.fs = frame_scores; c.seq = seq; c.n = n; c.fwd_prev = fwd_prev; c.rev_next = rev_next; c.quality = quality;
    c.frame = frame; c.stop_position = stop_position; c.prm = prm; c.ep = ep; c.starts = starts; c.cap = cap;
    for (i = 0; i < prm->n_start_codons; i++) c.pat[i] = orc_codon_from(prm->start_codon[i]);
    /* Score_Orfs_Errors (:1632-1646) */
    orc_score_orf_starts(&c, frame > 0 ? stop_position - 1 : stop_position + 3, 0, 0, NULL, NULL, 0);
    out->lo = c.lo0; out->hi = c.hi0; out->orf_is_truncated = c.trunc0;
    out->first_j = 0; out->accepted = 0; out->best_score = -1.7976931348623157e308;
    if (c.n_starts > cap) return c.n_starts;
    for (i = 0; i < c.n_starts; i++)
        if (starts[i].s.j > prm->ignore_score_len && 0.0 > starts[i].s.score) starts[i].s.score = 0.0;
    if (c.n_starts > 0) {
        /* first_j is the j of front() (forward) / back() (reverse) after std::sort by pos -- an unstable sort: when
         * several entries share the extreme pos with different j, only the real sort can tell (accepted = 2 if the
         * answer matters) */
        lo_pos = hi_pos = starts[0].s.pos;
        for (i = 1; i < c.n_starts; i++) {
            if (starts[i].s.pos < lo_pos) lo_pos = starts[i].s.pos;
            if (starts[i].s.pos > hi_pos) hi_pos = starts[i].s.pos;
        }
        {
            int jmin = 2147483647, jmax = -2147483647 - 1;
            for (i = 0; i < c.n_starts; i++)
                if (starts[i].s.pos == (frame > 0 ? lo_pos : hi_pos)) {
                    if (pick < 0) pick = i;
                    if (starts[i].s.j < jmin) jmin = starts[i].s.j;
                    if (starts[i].s.j > jmax) jmax = starts[i].s.j;
                }
            for (i = 0; i < c.n_starts; i++)
                if (starts[i].s.score > best) best = starts[i].s.score;
            out->first_j = jmin;                                            /* (the smallest of the candidates) */
            if (jmin + 1 >= prm->min_gene_len) amb = 0;                     /* whichever comes first passes */
            else if (jmax + 1 < prm->min_gene_len) amb = -1;                /* none passes */
            else amb = 1;
            if (amb >= 0) {
                out->best_score = best;
                out->accepted = best > prm->start_threshold ? (amb ? 2 : 1) : 0;
            }
        }
    }
    return c.n_starts;
}

/* Fasta_Read (fasta.cc:236-286): fgetc / ungetc restated as an index into a buffer */
int orc_fasta_next(const char *buf, long n, long *pos, long *hdr_begin, long *hdr_end, char *seq, long *seq_len)
{
    long i = *pos, k = 0;
    while (i < n && buf[i] != '>') i++;                 /* skip till next '>' if necessary */
    if (i >= n) { *pos = n; return 0; }
    i++;
    while (i < n && buf[i] == ' ') i++;                 /* skip spaces if any */
    if (i >= n) { *pos = n; return 0; }
    *hdr_begin = i;
    while (i < n && buf[i] != '\n') i++;                /* rest of line into hdr */
    *hdr_end = i;
    if (i < n) i++;
    while (i < n && buf[i] != '>') {                    /* everything up till next '>' into s */
        if (!isspace((unsigned char)buf[i])) seq[k++] = buf[i];
        i++;
    }
    *seq_len = k;
    *pos = i;                                           /* the '>' is pushed back */
    return 1;
}

long orc_fasta_all(const char *buf, long n, char *out, long *n_bases, long *gc)
{
    long pos = 0, hb, he, sl, recs = 0, total = 0, ct = 0, k;
    char *seq = (char *)malloc((size_t)n + 1);
    while (orc_fasta_next(buf, n, &pos, &hb, &he, seq, &sl)) {
        for (k = 0; k < sl; k++) {
            const int ch = tolower(orc_filter(seq[k]));    /* glimmer3.cc:270-271 */
            if (out) out[total + k] = (char)ch;
            ct += (ch == 'g' || ch == 'c');
        }
        total += sl;
        recs++;
    }
    free(seq);
    *n_bases = total; *gc = ct;
    return recs;
}

long orc_score_reads_6frame(const orc_model *gene, const orc_model *indep, const char *seqs,
                            int n_reads, int L, double *out)
{
    int r;
    for (r = 0; r < n_reads; r++)
        orc_score_all_frames(gene, indep, seqs + (size_t)r * L, L, out + (size_t)r * 6 * L);
    return (long)n_reads * L;
}

/* ------------------------------------------------------------------------ */
/* training: ICM_Training_t (src/ICM/icm.cc:1010-1455) and its helpers       */
/* ------------------------------------------------------------------------ */

/* icm.hh:37-43, 62-80 */
#define ORC_NUM_CHI2 7
static const float ORC_CHI2_VAL[ORC_NUM_CHI2] = {2.37, 4.11, 6.25, 7.81, 9.35, 11.3, 12.8};
static const float ORC_CHI2_SIG[ORC_NUM_CHI2] = {0.50, 0.75, 0.90, 0.95, 0.975, 0.99, 0.995};
static const double ORC_MUT_INFO_BIAS = 0.03;
static const double ORC_MUT_INFO_EPSILON = 1e-4;
static const double ORC_PSEUDO_COUNT = 0.001;
#define ORC_SAMPLE_SIZE_BOUND 400

static int orc_level_first(int level)   /* first node of a level: (4^level - 1) / 3 */
{
    int i, pw = 1;
    for (i = 0; i < level; i++) pw *= ORC_ALPHA;
    return (pw - 1) / (ORC_ALPHA - 1);
}

void orc_train_level_counts(const orc_model *m, const char *const *strings, int n_strings, int level,
                            int32_t *counts)
{
    const int W = m->model_len, P = m->periodicity;
    const int first = orc_level_first(level);
    const int on_level = orc_level_first(level + 1) - first;
    int s, i;

    if (level == 0) {
        /* icm.cc:1373-1390: one pass per sub-model, first window at `offset`, stepping by the periodicity
         * (Count_Char_Pairs, icm.cc:1841-1870) */
        int frame;
        for (frame = 0; frame < P; frame++) {
            int offset = frame - (W % P);
            if (offset < 0) offset += P;
            for (s = 0; s < n_strings; s++) {
                const char *str = strings[s];
                int len = (int)strlen(str), start, stop, end;
                if (offset >= len) continue;      /* the reference would read past the NUL here */
                str += offset;
                end = len - offset;
                for (start = 0, stop = W - 1; stop < end; start += P, stop += P) {
                    int last = orc_subscript(str[stop]);
                    if (W == 1)     /* no context: Count_Single_Chars (icm.cc:1874-1896), kept in table 0 at [last] */
                        counts[(size_t)frame * 16 + last]++;
                    for (i = 0; i < W - 1; i++)
                        counts[((size_t)frame * (W - 1) + i) * 16 + ORC_ALPHA * orc_subscript(str[start + i]) + last]++;
                }
            }
        }
        return;
    }

    /* icm.cc:1190-1229 with Get_Training_Node (icm.cc:1233-1256) inlined */
    for (s = 0; s < n_strings; s++) {
        const char *str = strings[s];
        int end = (int)strlen(str), start = 0, stop, frame = W % P;
        for (stop = W - 1; stop < end; start++, stop++) {
            const int16_t *mip = m->mip + (size_t)frame * m->num_nodes;
            int sub = 0, ok = 1;
            for (i = 0; i < level; i++) {
                int j = mip[sub];
                if (j < 0) { ok = 0; break; }
                sub = sub * ORC_ALPHA + orc_subscript(str[start + j]) + 1;
            }
            if (ok) {
                int last = orc_subscript(str[stop]);
                int32_t *ct = counts + ((size_t)frame * on_level + (sub - first)) * (W - 1) * 16;
                for (i = 0; i < W - 1; i++)
                    ct[i * 16 + ORC_ALPHA * orc_subscript(str[start + i]) + last]++;
            }
            if (++frame == P) frame = 0;
        }
    }
}

/* icm.cc:1900-1955 */
double orc_mutual_info(const int32_t ct[16], int sum)
{
    double mut_info = 0.0, left_prob[ORC_ALPHA], right_prob[ORC_ALPHA];
    int i, j, k;
    if (sum == 0) return 0.0;
    for (i = 0; i < ORC_ALPHA; i++) left_prob[i] = right_prob[i] = 0.0;
    for (i = k = 0; i < ORC_ALPHA; i++)
        for (j = 0; j < ORC_ALPHA; j++) {
            left_prob[i] += ct[k];
            right_prob[j] += ct[k];
            k++;
        }
    for (i = 0; i < ORC_ALPHA; i++) {
        left_prob[i] /= sum;
        right_prob[i] /= sum;
    }
    for (i = k = 0; i < ORC_ALPHA; i++)
        for (j = 0; j < ORC_ALPHA; j++) {
            double prob = (double)ct[k] / sum;
            if (prob != 0.0 && left_prob[i] != 0.0 && right_prob[j] != 0.0)
                mut_info += prob * log(prob / (left_prob[i] * right_prob[j]));
            k++;
        }
    return mut_info;
}

/* icm.cc:1260-1330.  prob / parent are the 4-float rows of the node and of its parent (plain probabilities). */
static void orc_interpolate_probs(float *prob, const float *parent, const int ct[ORC_ALPHA])
{
    double expected, chi2_stat, lambda, total_sum = 0.0;
    int i;
    for (i = 0; i < ORC_ALPHA; i++) total_sum += ct[i];
    for (i = 0; i < ORC_ALPHA; i++)
        prob[i] = (ct[i] + ORC_PSEUDO_COUNT * parent[i]) / (total_sum + ORC_PSEUDO_COUNT);
    if (total_sum >= ORC_SAMPLE_SIZE_BOUND) return;
    chi2_stat = 0.0;
    for (i = 0; i < ORC_ALPHA; i++) {
        expected = total_sum * parent[i];
        if (expected > 0.0) chi2_stat += pow(ct[i] - expected, 2.0) / expected;
    }
    for (i = 0; i < ORC_NUM_CHI2 && ORC_CHI2_VAL[i] < chi2_stat; i++)
        ;
    if (i == 0)
        lambda = 0.0;
    else if (i == ORC_NUM_CHI2)
        lambda = 1.0;
    else   /* the table differences are float subtractions, as in the reference */
        lambda = ORC_CHI2_SIG[i - 1]
                 + ((chi2_stat - ORC_CHI2_VAL[i - 1]) / (ORC_CHI2_VAL[i] - ORC_CHI2_VAL[i - 1]))
                       * (ORC_CHI2_SIG[i] - ORC_CHI2_SIG[i - 1]);
    lambda *= total_sum / ORC_SAMPLE_SIZE_BOUND;
    if (lambda > 1.0) lambda = 1.0;
    for (i = 0; i < ORC_ALPHA; i++) {
        prob[i] *= lambda;
        prob[i] += (1.0 - lambda) * parent[i];
    }
}

/* the max-mutual-information scan shared by the root (icm.cc:1406-1426) and the deeper nodes (icm.cc:1116-1139):
 * positions to the right win when within MUT_INFO_BIAS.  *used = info of the chosen position (deeper nodes),
 * *best = the maximum (root). */
static int orc_best_position(const int32_t *ct, int W, int sum, double *best, double *used)
{
    int i, max_pos = 0;
    double best_info = orc_mutual_info(ct, sum), next_info, used_info;
    used_info = best_info;
    for (i = 1; i < W - 1; i++) {
        next_info = orc_mutual_info(ct + 16 * i, sum);
        if (next_info >= best_info) {
            used_info = best_info = next_info;
            max_pos = i;
        } else if (next_info >= (best_info / (1.0 + ORC_MUT_INFO_BIAS))) {
            max_pos = i;
            used_info = next_info;
        }
    }
    *best = best_info;
    *used = used_info;
    return max_pos;
}

orc_model *orc_train_model(const char *const *strings, int n_strings, int model_len, int model_depth,
                           int periodicity, float *mut_info)
{
    orc_model *m = orc_model_new(model_len, model_depth, periodicity);
    const int W = model_len, P = periodicity;
    int frame, level, i, j, k, s;
    int32_t *counts = NULL;
    if (!m) return NULL;
    if (mut_info) memset(mut_info, 0, sizeof(float) * (size_t)P * m->num_nodes);

    if (model_depth == 0) {
        /* icm.cc:1376-1389: Count_Single_Chars (icm.cc:1874-1896) */
        for (frame = 0; frame < P; frame++) {
            int final_char_ct[ORC_ALPHA] = {0}, sum = 0, pos;
            float *prob = m->prob + 4 * (size_t)frame * m->num_nodes;
            int offset = frame - (W % P);
            if (offset < 0) offset += P;
            for (s = 0; s < n_strings; s++) {
                int len = (int)strlen(strings[s]);
                for (pos = offset + W - 1; pos < len; pos += P) final_char_ct[orc_subscript(strings[s][pos])]++;
            }
            for (i = 0; i < ORC_ALPHA; i++) sum += final_char_ct[i];
            for (i = 0; i < ORC_ALPHA; i++)
                prob[i] = (final_char_ct[i] + (float)(ORC_PSEUDO_COUNT / ORC_ALPHA)) / (sum + ORC_PSEUDO_COUNT);
            m->mip[(size_t)frame * m->num_nodes] = -1;
        }
    } else {
        /* roots: icm.cc:1391-1432 */
        counts = (int32_t *)calloc((size_t)P * (W - 1) * 16, sizeof(int32_t));
        orc_train_level_counts(m, strings, n_strings, 0, counts);
        for (frame = 0; frame < P; frame++) {
            const int32_t *ct = counts + (size_t)frame * (W - 1) * 16;
            int final_char_ct[ORC_ALPHA] = {0}, sum = 0, max_pos;
            float *prob = m->prob + 4 * (size_t)frame * m->num_nodes;
            double best, used;
            for (i = k = 0; i < ORC_ALPHA; i++)
                for (j = 0; j < ORC_ALPHA; j++) {
                    sum += ct[k];
                    final_char_ct[j] += ct[k];
                    k++;
                }
            for (j = 0; j < ORC_ALPHA; j++)   /* float arithmetic throughout, as written in the reference */
                prob[j] = (final_char_ct[j] + (float)(ORC_PSEUDO_COUNT / ORC_ALPHA)) / (float)(sum + ORC_PSEUDO_COUNT);
            max_pos = orc_best_position(ct, W, sum, &best, &used);
            m->mip[(size_t)frame * m->num_nodes] = (int16_t)max_pos;
            if (mut_info) mut_info[(size_t)frame * m->num_nodes] = (float)best;
        }
        free(counts);
    }

    /* Complete_Tree: icm.cc:1061-1186 */
    for (level = 1; level <= model_depth; level++) {
        const int first = orc_level_first(level), on_level = orc_level_first(level + 1) - first;
        counts = (int32_t *)calloc((size_t)P * on_level * (W - 1) * 16, sizeof(int32_t));
        orc_train_level_counts(m, strings, n_strings, level, counts);
        for (frame = 0; frame < P; frame++) {
            int16_t *mip = m->mip + (size_t)frame * m->num_nodes;
            float *probs = m->prob + 4 * (size_t)frame * m->num_nodes;
            int sub;
            for (sub = first; sub < first + on_level; sub++) {
                const int32_t *ct = counts + ((size_t)frame * on_level + (sub - first)) * (W - 1) * 16;
                int final_char_ct[ORC_ALPHA] = {0}, sum = 0, max_pos;
                double best, used;
                if (mip[orc_parent(sub)] < 0) {     /* stopped at the parent */
                    mip[sub] = -2;
                    continue;
                }
                for (i = k = 0; i < ORC_ALPHA; i++)
                    for (j = 0; j < ORC_ALPHA; j++) {
                        sum += ct[k];
                        final_char_ct[j] += ct[k];
                        k++;
                    }
                max_pos = orc_best_position(ct, W, sum, &best, &used);
                if (best <= ORC_MUT_INFO_EPSILON && sum < ORC_SAMPLE_SIZE_BOUND) max_pos = -1;
                mip[sub] = (int16_t)max_pos;
                if (mut_info) mut_info[(size_t)frame * m->num_nodes + sub] = (float)used;
                orc_interpolate_probs(probs + 4 * sub, probs + 4 * orc_parent(sub), final_char_ct);
            }
        }
        free(counts);
    }

    /* Take_Logs: icm.cc:1334-1352.  The argument is a float, so the C++ reference resolves  log  to the float
     * overload: logf, not log. */
    for (i = 0; i < 4 * P * m->num_nodes; i++)
        m->prob[i] = (m->prob[i] > 0.0) ? logf(m->prob[i]) : -FLT_MAX;
    return m;
}

/* ==== glimmer-mg -c: classification bookkeeping ==========================================================
 * Which ICM file scores which read, in which order, with which null-model GC and stop codons
 * (src/Glimmer/glimmer-mg.cc:326-375, 473-515, 726-758, 998-1027, 1211-1250, 1389-1420, 2050-2068, 2185-2219).
 *
 * The order comes from the iteration order of two __gnu_cxx::hash_map<string, ...> (glimmer-mg.hh:22-49 picks
 * <ext/hash_map>).  That container is not part of the reference tree: it is libstdc++'s SGI hash table
 * (GCC 11: /usr/include/c++/11/backward/hashtable.h, hash_fun.h; unchanged since GCC 3.1), whose published algorithm is
 * restated here:
 *   hash of a string        h = 5 h + c over its characters, unsigned long                (hash_fun.h __stl_hash_string)
 *   bucket counts           the next of 29 fixed primes >= the request; hash_map() asks for 100 -> 193
 *   operator[] / insert     resize (elements + 1) FIRST -- when that exceeds the bucket count every node moves to
 *                           the HEAD of its new bucket, old buckets in ascending order, chains front to back --, then
 *                           the key is looked up in bucket h % n and, if new, linked in at the HEAD of that bucket
 *   iteration               buckets in ascending order, each chain front to back
 * ========================================================================================================= */

static const unsigned long orc_sgi_primes[29] = {
    5ul, 53ul, 97ul, 193ul, 389ul, 769ul, 1543ul, 3079ul, 6151ul, 12289ul, 24593ul, 49157ul, 98317ul, 196613ul, 393241ul,
    786433ul, 1572869ul, 3145739ul, 6291469ul, 12582917ul, 25165843ul, 50331653ul, 100663319ul, 201326611ul, 402653189ul,
    805306457ul, 1610612741ul, 3221225473ul, 4294967291ul};

typedef struct orc_sgi_node { struct orc_sgi_node *next; char *key; void *val; } orc_sgi_node;
typedef struct orc_sgi_table { orc_sgi_node **bucket; unsigned long n_buckets, n_elements; } orc_sgi_table;

static unsigned long orc_sgi_next_prime(unsigned long n)
{
    int i;
    for (i = 0; i < 29; i++)
        if (orc_sgi_primes[i] >= n) return orc_sgi_primes[i];
    return orc_sgi_primes[28];
}

static unsigned long orc_sgi_hash(const char *s)
{
    unsigned long h = 0;
    for (; *s; ++s) h = 5 * h + (unsigned long)(long)*s;      /* `char` is signed on this platform, as in the reference's build */
    return h;
}

static void orc_sgi_init(orc_sgi_table *t)
{
    t->n_buckets = orc_sgi_next_prime(100);
    t->bucket = (orc_sgi_node **)calloc(t->n_buckets, sizeof(orc_sgi_node *));
    t->n_elements = 0;
}

static void orc_sgi_resize(orc_sgi_table *t, unsigned long hint)
{
    unsigned long n, b;
    orc_sgi_node **nb;
    if (hint <= t->n_buckets) return;
    n = orc_sgi_next_prime(hint);
    if (n <= t->n_buckets) return;
    nb = (orc_sgi_node **)calloc(n, sizeof(orc_sgi_node *));
    for (b = 0; b < t->n_buckets; b++) {
        orc_sgi_node *first = t->bucket[b];
        while (first) {
            const unsigned long k = orc_sgi_hash(first->key) % n;
            t->bucket[b] = first->next;
            first->next = nb[k];
            nb[k] = first;
            first = t->bucket[b];
        }
    }
    free(t->bucket);
    t->bucket = nb;
    t->n_buckets = n;
}

static orc_sgi_node *orc_sgi_find(const orc_sgi_table *t, const char *key)
{
    orc_sgi_node *c = t->bucket[orc_sgi_hash(key) % t->n_buckets];
    for (; c; c = c->next)
        if (strcmp(c->key, key) == 0) return c;
    return NULL;
}

/* hash_map::operator[]: find_or_insert of (key, empty value) */
static orc_sgi_node *orc_sgi_index(orc_sgi_table *t, const char *key)
{
    unsigned long k;
    orc_sgi_node *c;
    orc_sgi_resize(t, t->n_elements + 1);
    k = orc_sgi_hash(key) % t->n_buckets;
    for (c = t->bucket[k]; c; c = c->next)
        if (strcmp(c->key, key) == 0) return c;
    c = (orc_sgi_node *)calloc(1, sizeof *c);
    c->key = strdup(key);
    c->next = t->bucket[k];
    t->bucket[k] = c;
    t->n_elements++;
    return c;
}

static void orc_sgi_free(orc_sgi_table *t, void (*free_val)(void *))
{
    unsigned long b;
    for (b = 0; b < t->n_buckets; b++) {
        orc_sgi_node *c = t->bucket[b];
        while (c) {
            orc_sgi_node *nx = c->next;
            if (free_val && c->val) free_val(c->val);
            free(c->key);
            free(c);
            c = nx;
        }
    }
    free(t->bucket);
}

typedef struct orc_strlist { char **s; long n, cap; } orc_strlist;
static void orc_strlist_push(orc_strlist *l, const char *s)
{
    if (l->n == l->cap) {
        l->cap = l->cap ? 2 * l->cap : 4;
        l->s = (char **)realloc(l->s, (size_t)l->cap * sizeof(char *));
    }
    l->s[l->n++] = strdup(s);
}
static void orc_strlist_free(void *p)
{
    orc_strlist *l = (orc_strlist *)p;
    long i;
    for (i = 0; i < l->n; i++) free(l->s[i]);
    free(l->s);
    free(l);
}

struct orc_classes {
    char *icm_dir;
    orc_sgi_table classifications;      /* header prefix -> orc_strlist of classes          (glimmer-mg.cc:163-164) */
    orc_sgi_table icm_sequences;        /* ICM file -> orc_strlist of header prefixes       (:205-206) */
    orc_sgi_table gc;                   /* class -> float *                                 (:196-197) */
    orc_sgi_table transl;               /* class -> int *                                   (:199-200) */
};

/* split (s, '|') (src/Common/kelley.cc:10-26): fields 0 and 1 */
static int orc_strain_nc(const char *cls, char *strain, char *nc, size_t cap)
{
    const char *bar = strchr(cls, '|'), *bar2;
    size_t n;
    if (!bar) return -1;
    n = (size_t)(bar - cls);
    if (n >= cap) return -1;
    memcpy(strain, cls, n);
    strain[n] = 0;
    bar2 = strchr(bar + 1, '|');
    n = bar2 ? (size_t)(bar2 - bar - 1) : strlen(bar + 1);
    if (n >= cap) return -1;
    memcpy(nc, bar + 1, n);
    nc[n] = 0;
    return 0;
}

#include <sys/stat.h>

/* Classes_ICM_File (glimmer-mg.cc:473-515) */
static char *orc_classes_icm_name(const orc_classes *c, const orc_strlist *cl)
{
    char s1[512], n1[512], s2[512], n2[512];
    const size_t cap = strlen(c->icm_dir) + 2200;
    char *out = (char *)malloc(cap);
    long i;
    if (cl->n >= 2) {
        for (i = 1; i < cl->n; i++) {
            const int first_smaller = strcmp(cl->s[0], cl->s[i]) < 0;   /* string::compare: bytes as unsigned char, like strcmp */
            const char *a = first_smaller ? cl->s[0] : cl->s[i], *b = first_smaller ? cl->s[i] : cl->s[0];
            struct stat st;
            if (orc_strain_nc(a, s1, n1, sizeof s1) || orc_strain_nc(b, s2, n2, sizeof s2)) { free(out); return NULL; }
            snprintf(out, cap, "%s/%s/%s_2/%s/%s.gicm", c->icm_dir, s1, n1, s2, n2);
            if (stat(out, &st) == 0) return out;
        }
    }
    if (orc_strain_nc(cl->s[0], s1, n1, sizeof s1)) { free(out); return NULL; }
    snprintf(out, cap, "%s/%s/%s.gicm", c->icm_dir, s1, n1);
    return out;
}

void orc_classes_free(orc_classes *c)
{
    if (!c) return;
    orc_sgi_free(&c->classifications, orc_strlist_free);
    orc_sgi_free(&c->icm_sequences, orc_strlist_free);
    orc_sgi_free(&c->gc, free);
    orc_sgi_free(&c->transl, free);
    free(c->icm_dir);
    free(c);
}

orc_classes *orc_classes_load(const char *text, long n, const char *icm_dir)
{
    orc_classes *c = (orc_classes *)calloc(1, sizeof *c);
    long pos = 0;
    unsigned long b;
    c->icm_dir = strdup(icm_dir);
    orc_sgi_init(&c->classifications);
    orc_sgi_init(&c->icm_sequences);
    orc_sgi_init(&c->gc);
    orc_sgi_init(&c->transl);
    /* Parse_Classes (:726-758): getline, split on white space (kelley.cc:34-53), classifications[a[0]] = a[1..] */
    while (pos < n) {
        long end = pos, i;
        orc_strlist *v = (orc_strlist *)calloc(1, sizeof *v);
        char *key = NULL;
        while (end < n && text[end] != '\n') end++;
        i = pos;
        while (i < end) {
            long b0;
            while (i < end && (text[i] == ' ' || text[i] == '\t' || text[i] == '\r')) i++;
            b0 = i;
            while (i < end && !(text[i] == ' ' || text[i] == '\t' || text[i] == '\r')) i++;
            if (i > b0) {
                char *tok = (char *)malloc((size_t)(i - b0) + 1);
                memcpy(tok, text + b0, (size_t)(i - b0));
                tok[i - b0] = 0;
                if (!key) key = tok;
                else { orc_strlist_push(v, tok); free(tok); }
            }
        }
        pos = end + 1;
        if (!key || v->n == 0) {        /* undefined behaviour in the reference (a[0] of an empty vector, Seq_Classes[0]) */
            free(key);
            orc_strlist_free(v);
            orc_classes_free(c);
            return NULL;
        }
        {
            orc_sgi_node *nd = orc_sgi_index(&c->classifications, key);
            if (nd->val) orc_strlist_free(nd->val);     /* a later line for the same read replaces the earlier one */
            nd->val = v;
        }
        free(key);
    }
    /* Read_Meta_ICMs (:998-1027), Read_Meta_GC (:1389-1420), Read_Meta_Stops (:1211-1250): the classifications in the
     * table's own order */
    for (b = 0; b < c->classifications.n_buckets; b++) {
        orc_sgi_node *ci;
        for (ci = c->classifications.bucket[b]; ci; ci = ci->next) {
            const orc_strlist *cl = (const orc_strlist *)ci->val;
            char *icm_file = orc_classes_icm_name(c, cl);
            char strain[512], nc[512], path[4096], line[4096];
            orc_sgi_node *isi;
            long i;
            if (!icm_file) { orc_classes_free(c); return NULL; }
            isi = orc_sgi_find(&c->icm_sequences, icm_file);
            if (!isi) {
                isi = orc_sgi_index(&c->icm_sequences, icm_file);
                isi->val = calloc(1, sizeof(orc_strlist));
            }
            orc_strlist_push((orc_strlist *)isi->val, ci->key);
            free(icm_file);
            for (i = 0; i < cl->n; i++) {
                FILE *fp;
                float *g;
                if (orc_sgi_find(&c->gc, cl->s[i])) continue;
                if (orc_strain_nc(cl->s[i], strain, nc, sizeof strain)) { orc_classes_free(c); return NULL; }
                snprintf(path, sizeof path, "%s/%s/%s.gc.txt", c->icm_dir, strain, nc);
                g = (float *)malloc(sizeof(float));
                fp = fopen(path, "r");
                if (fp) {
                    line[0] = 0;
                    if (!fgets(line, sizeof line, fp)) line[0] = 0;
                    *g = (float)strtod(line, NULL);     /* Sequence_GC holds floats */
                    fclose(fp);
                } else
                    *g = 0.5f;                          /* "WARNING: GC classification file unavailable" */
                orc_sgi_index(&c->gc, cl->s[i])->val = g;
            }
            if (!orc_sgi_find(&c->transl, cl->s[0])) {
                FILE *fp;
                int *code = (int *)malloc(sizeof(int));
                *code = 11;
                orc_strain_nc(cl->s[0], strain, nc, sizeof strain);
                snprintf(path, sizeof path, "%s/%s/%s.gbk", c->icm_dir, strain, nc);
                fp = fopen(path, "r");
                if (fp) {
                    while (fgets(line, sizeof line, fp)) {
                        const char *tt = strstr(line, "transl_table=");
                        if (tt) { *code = (int)strtol(tt + 13, NULL, 10); break; }
                    }
                    fclose(fp);
                }
                orc_sgi_index(&c->transl, cl->s[0])->val = code;
            }
        }
    }
    return c;
}

int orc_classes_n_icms(const orc_classes *c) { return (int)c->icm_sequences.n_elements; }

/* ICM file k in the iteration order of ICM_Sequences (the loop of glimmer-mg.cc:361) */
const char *orc_classes_icm_file(const orc_classes *c, int k)
{
    unsigned long b;
    for (b = 0; b < c->icm_sequences.n_buckets; b++) {
        const orc_sgi_node *nd;
        for (nd = c->icm_sequences.bucket[b]; nd; nd = nd->next)
            if (k-- == 0) return nd->key;
    }
    return NULL;
}

/* One chunk of the main loop (glimmer-mg.cc:326-375): Read_Indexes[prefix] = index for every read of the chunk, then ICM by
 * ICM, read by read.  hdr: NUL-terminated header lines.  Returns the number of reads processed. */
long orc_classes_plan(const orc_classes *c, const char *const *hdr, long n, long *order, long *icm_begin, double *gc,
                      int *transl)
{
    orc_sgi_table read_index;
    long i, k = 0, f = 0;
    unsigned long b;
    orc_sgi_init(&read_index);
    for (i = 0; i < n; i++) {
        const char *h = hdr[i];
        char *key;
        size_t b0 = 0, e;
        while (h[b0] == ' ' || h[b0] == '\t' || h[b0] == '\n' || h[b0] == '\r') b0++;
        e = b0;
        while (h[e] && !(h[e] == ' ' || h[e] == '\t' || h[e] == '\n' || h[e] == '\r')) e++;
        if (e == b0) continue;                          /* split (hdr)[0] of an empty vector: undefined in the reference */
        key = (char *)malloc(e - b0 + 1);
        memcpy(key, h + b0, e - b0);
        key[e - b0] = 0;
        orc_sgi_index(&read_index, key)->val = (void *)(size_t)(i + 1);     /* a later read of the same key wins */
        free(key);
    }
    for (b = 0; b < c->icm_sequences.n_buckets; b++) {
        const orc_sgi_node *nd;
        for (nd = c->icm_sequences.bucket[b]; nd; nd = nd->next) {
            const orc_strlist *reads = (const orc_strlist *)nd->val;
            long r;
            icm_begin[f++] = k;
            for (r = 0; r < reads->n; r++) {
                const orc_sgi_node *ri = orc_sgi_find(&read_index, reads->s[r]);
                const orc_strlist *cl;
                if (!ri) continue;
                order[k] = (long)(size_t)ri->val - 1;
                cl = (const orc_strlist *)orc_sgi_find(&c->classifications, reads->s[r])->val;
                if (gc) {               /* Update_Meta_Null_ICM (:2058-2064) */
                    const float num_classes = (float)cl->n;
                    double g = 0.0;
                    unsigned int s;
                    for (s = 0; s < num_classes; s++) g += *(const float *)orc_sgi_find(&c->gc, cl->s[s])->val;
                    g /= num_classes;
                    gc[k] = g;
                }
                if (transl) transl[k] = *(const int *)orc_sgi_find(&c->transl, cl->s[0])->val;   /* Update_Meta_Stop (:2196) */
                k++;
            }
        }
    }
    icm_begin[f] = k;
    orc_sgi_free(&read_index, NULL);
    return k;
}

/* Set_Stop_Codons_By_Code (src/Common/gene.cc:1560-1624); returns the number of codons, 0 for an unknown table */
int orc_stop_codons_by_code(int code, const char *out[8])
{
    int n = 0;
    switch (code) {
    case 1: case 11: case 12: out[n++] = "taa"; out[n++] = "tag"; out[n++] = "tga"; break;
    case 2: out[n++] = "taa"; out[n++] = "tag"; out[n++] = "aga"; out[n++] = "agg"; break;
    case 3: case 4: case 5: case 9: case 10: case 13: case 21: out[n++] = "taa"; out[n++] = "tag"; break;
    case 6: out[n++] = "tga"; break;
    case 14: out[n++] = "tag"; break;
    case 15: case 16: out[n++] = "taa"; out[n++] = "tga"; break;
    case 22: out[n++] = "taa"; out[n++] = "tga"; out[n++] = "tca"; break;
    case 23: out[n++] = "taa"; out[n++] = "tag"; out[n++] = "tga"; out[n++] = "tta"; break;
    default: break;
    }
    return n;
}

/* Set_Ignore_Score_Len (src/Glimmer/glimmer_base.cc:2597-2633) */
int orc_ignore_score_len(double gc_frac, const char *const *stop_codon, int n_stops)
{
    double poisson_lambda = 0.0;
    int i, j;
    for (i = 0; i < n_stops; i++) {
        double x = 1.0;
        for (j = 0; j < 3; j++)
            if (stop_codon[i][j] == 'c' || stop_codon[i][j] == 'g') x *= gc_frac / 2.0;
            else x *= (1.0 - gc_frac) / 2.0;
        poisson_lambda += x;
    }
    return (int)(long)floor(3.0 * log(2.0 * 1000000 * poisson_lambda) / poisson_lambda);
}
