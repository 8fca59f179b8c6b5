/*
 * bwt_oracle.c -- CPU restatement of the reference BWT block codec.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the GPU path is compared
 * against (tests/, the smoke check and the benchmark's cpu_baseline leg).  Nothing
 * in the shipped library or the host layer may call, link or load it.
 *
 * Parity status: PINNED.  Every function below is checked byte-for-byte against
 * the unmodified reference compiled from its own sources (oracle/_ref, built by
 * oracle/Makefile) and against the reference's known-answer vectors
 * (tests/test_small.cpp:79-93) by tests/test_oracle.py.
 *
 * Each function names the reference lines it restates (paths are relative to
 * the reference checkout).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* Delimiter search: src/bwt.cpp:19-47 (find_unique_char).  The reference
 * streams the file in 8 KB chunks into a set of seen bytes and returns the
 * lowest byte value 0..255 that never occurs, or -1 when all 256 occur. */
int oracle_find_unique_byte(const uint8_t *data, size_t len)
{
    unsigned char seen[256];
    memset(seen, 0, sizeof seen);
    for (size_t i = 0; i < len; ++i)
        seen[data[i]] = 1;
    for (int b = 0; b < 256; ++b)
        if (!seen[b])
            return b;
    return -1;
}

/* ------------------------------------------------------------------------- */
/* Suffix array by prefix doubling: src/bwt.cpp:50-99 (build_suffix_array).
 *
 *   rank[i] = unsigned byte value                         (:58-61)
 *   for k = 1, 2, 4, ... < n:                             (:64)
 *     sort suffix_array by (rank[a], rank[a+k]) where a missing second
 *     component (a+k >= n) orders before every present one   (:66-82)
 *     temp[sa[0]] = 0; temp[sa[i]] = temp[sa[i-1]] + (cmp(sa[i-1], sa[i]) ? 1 : 0)   (:85-88)
 *     rank = temp; stop when rank[sa[n-1]] == n-1          (:90-95)
 *
 * The comparator is a strict weak order, so any comparison sort yields the
 * same equivalence classes; the final array is unique because all suffixes of
 * a string are distinct once the "shorter is smaller" rule is applied.  The
 * sort below is an introsort over size_t with the comparator inlined (the
 * reference uses std::sort, same algorithm class).
 */
typedef struct {
    const size_t *rank;
    size_t k, n;
} sa_cmp_ctx;

static inline int sa_less(const sa_cmp_ctx *c, size_t a, size_t b)
{
    size_t ra = c->rank[a], rb = c->rank[b];
    if (ra != rb)
        return ra < rb;
    int a_has = (a + c->k < c->n);
    int b_has = (b + c->k < c->n);
    if (a_has && b_has)
        return c->rank[a + c->k] < c->rank[b + c->k];
    return !a_has && b_has;
}

static void sa_insertion(size_t *v, size_t lo, size_t hi, const sa_cmp_ctx *c)
{
    for (size_t i = lo + 1; i < hi; ++i) {
        size_t x = v[i], j = i;
        while (j > lo && sa_less(c, x, v[j - 1])) {
            v[j] = v[j - 1];
            --j;
        }
        v[j] = x;
    }
}

static void sa_sift(size_t *v, size_t start, size_t end, size_t base, const sa_cmp_ctx *c)
{
    size_t root = start;
    for (;;) {
        size_t child = 2 * root + 1;
        if (child >= end)
            break;
        if (child + 1 < end && sa_less(c, v[base + child], v[base + child + 1]))
            ++child;
        if (!sa_less(c, v[base + root], v[base + child]))
            break;
        size_t t = v[base + root];
        v[base + root] = v[base + child];
        v[base + child] = t;
        root = child;
    }
}

static void sa_heapsort(size_t *v, size_t lo, size_t hi, const sa_cmp_ctx *c)
{
    size_t m = hi - lo;
    for (size_t s = m / 2; s-- > 0;)
        sa_sift(v, s, m, lo, c);
    for (size_t e = m; e-- > 1;) {
        size_t t = v[lo];
        v[lo] = v[lo + e];
        v[lo + e] = t;
        sa_sift(v, 0, e, lo, c);
    }
}

static void sa_introsort(size_t *v, size_t lo, size_t hi, int depth, const sa_cmp_ctx *c)
{
    while (hi - lo > 16) {
        if (depth-- == 0) {
            sa_heapsort(v, lo, hi, c);
            return;
        }
        /* median of three, moved to v[lo] so the Hoare scan below always
         * leaves both sides non-empty */
        size_t mid = lo + (hi - lo) / 2, a = v[lo], b = v[mid], d = v[hi - 1], pi;
        if (sa_less(c, a, b))
            pi = sa_less(c, b, d) ? mid : (sa_less(c, a, d) ? hi - 1 : lo);
        else
            pi = sa_less(c, a, d) ? lo : (sa_less(c, b, d) ? hi - 1 : mid);
        size_t p = v[pi];
        v[pi] = v[lo];
        v[lo] = p;
        size_t i = lo, j = hi - 1;
        for (;;) {
            while (sa_less(c, v[i], p))
                ++i;
            while (sa_less(c, p, v[j]))
                --j;
            if (i >= j)
                break;
            size_t t = v[i];
            v[i] = v[j];
            v[j] = t;
            ++i;
            --j;
        }
        /* [lo, j] <= p <= [j+1, hi) ; recurse on the smaller side */
        if (j + 1 - lo < hi - (j + 1)) {
            sa_introsort(v, lo, j + 1, depth, c);
            lo = j + 1;
        } else {
            sa_introsort(v, j + 1, hi, depth, c);
            hi = j + 1;
        }
    }
    sa_insertion(v, lo, hi, c);
}

/* Returns the number of doubling rounds executed (diagnostic). */
int oracle_build_suffix_array(const uint8_t *s, size_t n, size_t *sa)
{
    if (n == 0)
        return 0;
    size_t *rank = (size_t *)malloc(n * sizeof(size_t));
    size_t *temp = (size_t *)malloc(n * sizeof(size_t));
    int rounds = 0;
    for (size_t i = 0; i < n; ++i) {
        sa[i] = i;
        rank[i] = s[i];
    }
    int depth = 2;
    for (size_t m = n; m > 1; m >>= 1)
        depth += 2;
    for (size_t k = 1; k < n; k *= 2) {
        sa_cmp_ctx c = {rank, k, n};
        sa_introsort(sa, 0, n, depth, &c);
        ++rounds;
        temp[sa[0]] = 0;
        for (size_t i = 1; i < n; ++i)
            temp[sa[i]] = temp[sa[i - 1]] + (sa_less(&c, sa[i - 1], sa[i]) ? 1 : 0);
        memcpy(rank, temp, n * sizeof(size_t));
        if (rank[sa[n - 1]] == n - 1)
            break;
    }
    free(rank);
    free(temp);
    return rounds;
}

/* ------------------------------------------------------------------------- */
/* Forward transform of one block: src/bwt.cpp:102-123 (bwt_forward).
 * s = input + delimiter; out[i] = s[sa[i]-1], or s[n-1] when sa[i] == 0.
 * `out` receives len+1 bytes.  Returns the number of doubling rounds. */
int oracle_bwt_forward(const uint8_t *in, size_t len, uint8_t delimiter, uint8_t *out)
{
    size_t n = len + 1;
    uint8_t *s = (uint8_t *)malloc(n);
    size_t *sa = (size_t *)malloc(n * sizeof(size_t));
    memcpy(s, in, len);
    s[len] = delimiter;
    int rounds = oracle_build_suffix_array(s, n, sa);
    for (size_t i = 0; i < n; ++i)
        out[i] = (sa[i] == 0) ? s[n - 1] : s[sa[i] - 1];
    free(sa);
    free(s);
    return rounds;
}

/* ------------------------------------------------------------------------- */
/* Inverse transform of one block: src/inverse_bwt.cpp:13-63 (bwt_inverse).
 *   pass 1 (:23-32): occ[i] = number of L[i] in L[0..i); delimiter_row = LAST i with L[i] == delimiter
 *   C[]    (:35-45): exclusive prefix sum of symbol counts in unsigned byte order
 *   walk   (:48-59): row = delimiter_row; loop { row = C[L[row]] + occ[row]; ch = L[row];
 *                    stop at the first ch == delimiter; else append ch }
 *   reverse (:61-62)
 * `out` must hold n bytes.  Returns the output length (n-1 for a well-formed
 * block), or -1 if the block holds no delimiter (the reference indexes out of
 * bounds there; an error is returned instead). */
long oracle_bwt_inverse(const uint8_t *L, size_t n, uint8_t delimiter, uint8_t *out)
{
    if (n == 0)
        return -1;
    size_t *occ = (size_t *)malloc(n * sizeof(size_t));
    size_t count[256], first[256];
    memset(count, 0, sizeof count);
    size_t delimiter_row = (size_t)-1;
    for (size_t i = 0; i < n; ++i) {
        uint8_t ch = L[i];
        if (ch == delimiter)
            delimiter_row = i;
        occ[i] = count[ch]++;
    }
    if (delimiter_row == (size_t)-1) {
        free(occ);
        return -1;
    }
    size_t total = 0;
    for (int c = 0; c < 256; ++c) {
        first[c] = total;
        total += count[c];
    }
    size_t m = 0, row = delimiter_row;
    for (;;) {
        row = first[L[row]] + occ[row];
        uint8_t ch = L[row];
        if (ch == delimiter)
            break;
        out[m++] = ch;
    }
    for (size_t i = 0, j = m; i + 1 < j; ++i) {
        --j;
        uint8_t t = out[i];
        out[i] = out[j];
        out[j] = t;
    }
    free(occ);
    return (long)m;
}

/* ------------------------------------------------------------------------- */
/* Whole-buffer framing, forward: src/bwt.cpp:159-237 (bwt_forward_process_file),
 * writer order :131-140.  Layout: [delimiter][block 0: min(bs,rem)+1 bytes][block 1]...
 * The reference runs hardware_concurrency() workers over independent blocks
 * (:178-201); `nthreads` workers do the same here, each writing its block at
 * the closed-form offset 1 + i*(bs+1).  Returns bytes written to `out`
 * (len + 1 + ceil(len/bs)), or -1 when no byte value is free (:163-166). */
typedef struct {
    const uint8_t *in;
    uint8_t *out;
    size_t len, bs, nblocks;
    uint8_t delim;
    size_t next; /* guarded by mu */
    pthread_mutex_t mu;
} fwd_job;

static void *fwd_worker(void *p)
{
    fwd_job *j = (fwd_job *)p;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        size_t b = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (b >= j->nblocks)
            return NULL;
        size_t off = b * j->bs;
        size_t l = j->len - off < j->bs ? j->len - off : j->bs;
        oracle_bwt_forward(j->in + off, l, j->delim, j->out + 1 + b * (j->bs + 1));
    }
}

long oracle_forward_buffer(const uint8_t *in, size_t len, size_t block_size, uint8_t *out, int nthreads)
{
    int d = oracle_find_unique_byte(in, len);
    if (d < 0 || block_size == 0)
        return -1;
    size_t nblocks = (len + block_size - 1) / block_size;
    out[0] = (uint8_t)d;
    fwd_job j;
    j.in = in;
    j.out = out;
    j.len = len;
    j.bs = block_size;
    j.nblocks = nblocks;
    j.delim = (uint8_t)d;
    j.next = 0;
    pthread_mutex_init(&j.mu, NULL);
    if (nthreads < 1)
        nthreads = 1;
    if ((size_t)nthreads > nblocks)
        nthreads = (int)(nblocks ? nblocks : 1);
    pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int i = 1; i < nthreads; ++i)
        pthread_create(&t[i], NULL, fwd_worker, &j);
    fwd_worker(&j);
    for (int i = 1; i < nthreads; ++i)
        pthread_join(t[i], NULL);
    free(t);
    pthread_mutex_destroy(&j.mu);
    return (long)(len + 1 + nblocks);
}

/* Whole-buffer framing, inverse: src/inverse_bwt.cpp:66-99
 * (bwt_inverse_process_file).  First byte = delimiter (:77-82), then chunks of
 * block_size+1 bytes, the last one shorter (:85-95); serial like the reference.
 * Returns bytes written, -1 for an empty input (:78-82), -2 for a chunk with no
 * delimiter. */
long oracle_inverse_buffer(const uint8_t *in, size_t len, size_t block_size, uint8_t *out)
{
    if (len == 0 || block_size == 0)
        return -1;
    uint8_t delim = in[0];
    size_t pos = 1, w = 0, chunk = block_size + 1;
    while (pos < len) {
        size_t l = len - pos < chunk ? len - pos : chunk;
        long r = oracle_bwt_inverse(in + pos, l, delim, out + w);
        if (r < 0)
            return -2;
        w += (size_t)r;
        pos += l;
    }
    return (long)w;
}

/* ------------------------------------------------------------------------- */
/* File-to-file drivers, same return codes as the reference (0 ok, 1 error). */
static uint8_t *slurp(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc(sz > 0 ? (size_t)sz : 1);
    size_t got = sz > 0 ? fread(buf, 1, (size_t)sz, f) : 0;
    fclose(f);
    *len = got;
    return buf;
}

static int spill(const char *path, const uint8_t *buf, size_t len)
{
    FILE *f = fopen(path, "wb");
    if (!f)
        return 1;
    if (len)
        fwrite(buf, 1, len, f);
    fclose(f);
    return 0;
}

int oracle_forward_file(const char *in_path, const char *out_path, size_t block_size, int nthreads)
{
    size_t len = 0;
    uint8_t *in = slurp(in_path, &len);
    if (!in)
        return 1;
    size_t nblocks = block_size ? (len + block_size - 1) / block_size : 0;
    uint8_t *out = (uint8_t *)malloc(len + 1 + nblocks);
    long r = oracle_forward_buffer(in, len, block_size, out, nthreads);
    int rc = 1;
    if (r >= 0)
        rc = spill(out_path, out, (size_t)r);
    free(in);
    free(out);
    return rc;
}

int oracle_inverse_file(const char *in_path, const char *out_path, size_t block_size)
{
    size_t len = 0;
    uint8_t *in = slurp(in_path, &len);
    if (!in)
        return 1;
    uint8_t *out = (uint8_t *)malloc(len ? len : 1);
    long r = oracle_inverse_buffer(in, len, block_size, out);
    int rc = 1;
    if (r >= 0)
        rc = spill(out_path, out, (size_t)r);
    free(in);
    free(out);
    return rc;
}
