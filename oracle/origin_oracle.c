/*
 * origin_oracle.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the origin-pointer block format
 * of the reference's bzip2 comparison tool.
 *
 *   container  reference util/bzip2_bwt_utils.cpp:137-155 (write), :157-198 (read): per block one
 *              marker byte 0xFF, the origin row as three bytes big-endian, then the block's last
 *              column (block_size bytes, the file's last block possibly shorter);
 *   inverse    :200-253: Occ(i) = occurrences of L[i] before i, C(c) = rows whose first byte is
 *              below c, then len steps { emit L[row]; row = C(L[row]) + Occ(row) } from the origin
 *              row, the emitted bytes reversed;
 *   file level :316-349: blocks until the input ends or a block fails to parse (what was written so
 *              far stays, return 0).
 *
 * The forward half of that tool calls BZ2_blockSort (bzip2 1.0.x, an un-vendored submodule that is
 * absent from the reference checkout), so there is neither a reference build nor a reference-held
 * vector for this format: PARITY UNPINNED.  oracle_origin_forward below is the textbook definition
 * (sort the rotations of the block, last column, row of rotation 0) used to make test inputs; among
 * equal rotations (a block that is a shorter string repeated) any order gives the same last column,
 * and the row reported is the first of the tied ones -- every choice inverts to the same text.
 *
 * Only tests/ may load this file's library.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const uint8_t *s;
    size_t n;
} rot_ctx;

static const rot_ctx *g_rot;  /* single-threaded helper: qsort has no context argument */

static int rot_cmp(const void *pa, const void *pb)
{
    const size_t a = *(const size_t *)pa, b = *(const size_t *)pb;
    const uint8_t *s = g_rot->s;
    const size_t n = g_rot->n;
    for (size_t k = 0; k < n; ++k) {
        const uint8_t ca = s[(a + k) % n], cb = s[(b + k) % n];
        if (ca != cb) return ca < cb ? -1 : 1;
    }
    return a < b ? -1 : a > b ? 1 : 0;  /* equal rotations: by position, so the result is deterministic */
}

/* Last column of the sorted rotations of s[0..n) into out[0..n); returns the row of rotation 0. */
long oracle_origin_forward(const uint8_t *s, size_t n, uint8_t *out)
{
    if (n == 0) return 0;
    size_t *rot = malloc(n * sizeof *rot);
    if (!rot) return -1;
    for (size_t i = 0; i < n; ++i) rot[i] = i;
    rot_ctx c = {s, n};
    g_rot = &c;
    qsort(rot, n, sizeof *rot, rot_cmp);
    long orig = -1;
    for (size_t i = 0; i < n; ++i) {
        out[i] = s[(rot[i] + n - 1) % n];
        if (rot[i] == 0) orig = (long)i;
    }
    free(rot);
    return orig;
}

/* reference util/bzip2_bwt_utils.cpp:200-253 */
int oracle_origin_inverse(const uint8_t *L, size_t len, size_t orig, uint8_t *out)
{
    if (len == 0) return 0;
    if (orig >= len) return 1;
    size_t *occ = malloc(len * sizeof *occ);
    if (!occ) return 1;
    size_t count[256] = {0}, first[256];
    for (size_t i = 0; i < len; ++i) occ[i] = count[L[i]]++;
    size_t total = 0;
    for (int c = 0; c < 256; ++c) {
        first[c] = total;
        total += count[c];
    }
    size_t row = orig;
    for (size_t i = 0; i < len; ++i) {
        const uint8_t ch = L[row];
        out[len - 1 - i] = ch;
        row = first[ch] + occ[row];
    }
    free(occ);
    return 0;
}

/* Container of n text bytes in blocks of block_size: writes 4 + len bytes per block; returns the
 * bytes written (out must hold n + 4 * blocks). */
long oracle_origin_forward_buffer(const uint8_t *in, size_t n, size_t block_size, uint8_t *out)
{
    size_t w = 0;
    for (size_t off = 0; off < n; off += block_size) {
        const size_t len = n - off < block_size ? n - off : block_size;
        const long orig = oracle_origin_forward(in + off, len, out + w + 4);
        if (orig < 0) return -1;
        out[w] = 0xFF;                        /* :140-142 */
        out[w + 1] = (uint8_t)(orig >> 16);   /* :145-149 */
        out[w + 2] = (uint8_t)(orig >> 8);
        out[w + 3] = (uint8_t)orig;
        w += 4 + len;
    }
    return (long)w;
}

/* The reading loop of :316-349 over a buffer; returns the text bytes produced. */
long oracle_origin_inverse_buffer(const uint8_t *in, size_t n, size_t block_size, uint8_t *out)
{
    size_t r = 0, w = 0;
    while (r < n) {
        if (in[r] != 0xFF) break;             /* :164-167 invalid marker: stop */
        if (n - r < 4) break;                 /* :171-175 */
        const size_t orig = ((size_t)in[r + 1] << 16) | ((size_t)in[r + 2] << 8) | in[r + 3];
        const size_t len = n - r - 4 < block_size ? n - r - 4 : block_size;
        if (len == 0) break;                  /* :183 */
        if (orig >= len) break;               /* :189-192 */
        if (oracle_origin_inverse(in + r + 4, len, orig, out + w)) return -1;
        r += 4 + len;
        w += len;
    }
    return (long)w;
}
