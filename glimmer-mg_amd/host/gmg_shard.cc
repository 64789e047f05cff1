// gmg_shard.cc -- how one job is cut over the GPUs of a node and, inside one GPU's share, into batches (host only).
//
// glimmer3 / glimmer-mg treat every sequence on its own (src/Glimmer/glimmer3.cc:262-310, glimmer-mg.cc:361-451); the
// only state shared by the reads of a run is the GC fraction of the null model, a ratio of two counts over all
// bases (Set_GC_Fraction, src/Glimmer/glimmer_base.cc:2564-2595).  So a job shards over reads: contiguous ranges of
// about equal BASE count, one process per GPU, two integers per shard summed on the host, results concatenated in
// shard order -- no collective in the data path.

#include "../../include/gmg.h"

#include <stdint.h>
#include <stddef.h>

extern int gmg_set_error(int code, const char *fmt, ...);

// Cut k lies at the read boundary nearest to k * total / n_shards (ties: the earlier one), and the cuts never
// cross: every base belongs to exactly one shard, shards are contiguous, and a shard differs from total / n_shards
// by at most one read on either side.
extern "C" int gmg_shard_plan(const uint64_t *base_offsets, uint64_t n_reads, int n_shards, uint64_t *read_begin)
{
    if (!base_offsets || !read_begin || n_shards < 1) return gmg_set_error(GMG_EINVAL, "gmg_shard_plan: bad argument");
    const uint64_t total = base_offsets[n_reads] - base_offsets[0];
    read_begin[0] = 0;
    uint64_t r = 0;
    for (int k = 1; k < n_shards; k++) {
        // want = base_offsets[0] + round (k * total / n_shards) without overflow of 64 bits for totals < 2^57
        const unsigned __int128 num = (unsigned __int128)total * (unsigned)k;
        const uint64_t want = base_offsets[0] + (uint64_t)(num / (unsigned)n_shards);
        uint64_t lo = r, hi = n_reads;                   // first read boundary at or behind `want`, not before the previous cut
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (base_offsets[mid] < want) lo = mid + 1; else hi = mid;
        }
        if (lo > r && want - base_offsets[lo - 1] <= base_offsets[lo] - want) lo--;   // the boundary in front of it is nearer
        r = lo;
        read_begin[k] = r;
    }
    read_begin[n_shards] = n_reads;
    return GMG_OK;
}

// The same for a FASTA file that nobody has parsed yet: byte ranges cut where a record certainly starts (a '>' right
// behind a newline, as gmg_fasta_split), nearest behind k * n_bytes / n_shards.  Bytes are the measure there -- with
// headers of similar length that is the base count up to a constant.  cuts[0] = 0 .. cuts[n_shards] = n_bytes; a
// shard may be empty (fewer records than shards).
extern "C" int gmg_fasta_shard_ranges(const char *bytes, uint64_t n_bytes, int n_shards, uint64_t *cuts)
{
    if ((!bytes && n_bytes) || !cuts || n_shards < 1) return gmg_set_error(GMG_EINVAL, "gmg_fasta_shard_ranges: bad argument");
    cuts[0] = 0;
    for (int k = 1; k < n_shards; k++) {
        uint64_t i = (uint64_t)(((unsigned __int128)n_bytes * (unsigned)k) / (unsigned)n_shards);
        if (i < cuts[k - 1]) i = cuts[k - 1];
        if (i == 0) i = 1;
        while (i < n_bytes && !(bytes[i] == '>' && bytes[i - 1] == '\n')) i++;
        cuts[k] = i < n_bytes ? i : n_bytes;
    }
    cuts[n_shards] = n_bytes;
    return GMG_OK;
}

// Indep_GC_Frac of the whole job from the per-shard counts gmg_fasta_info returns (glimmer_base.cc:2564-2595: the
// ratio of two integers over all sequences of the file).  The reference counts in `unsigned int`: beyond 2^32 bases
// both of its counters wrap; as_reference != 0 reproduces that (byte-identical output on such files), 0 gives the
// fraction the author meant.
extern "C" double gmg_gc_fraction(const uint64_t *gc_counts, const uint64_t *base_counts, int n_shards, int as_reference)
{
    uint64_t gc = 0, total = 0;
    for (int k = 0; k < n_shards; k++) { gc += gc_counts[k]; total += base_counts[k]; }
    if (as_reference) return double((unsigned int)gc) / (unsigned int)total;        // (0 / 0 = NaN there as well)
    return total ? double(gc) / double(total) : 0.0;
}
