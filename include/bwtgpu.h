/*
 * bwtgpu.h -- C ABI of the MI355X (gfx950) BWT block codec.
 *
 * This is the drop-in boundary: the C++ host layer (bwt_amd/host/) implements the
 * reference's public API -- bwt_forward / bwt_inverse / bwt_*_process_file /
 * FileProcessor (reference src/bwt.hpp:9,14, src/inverse_bwt.hpp:8,13,
 * src/file_processor.hpp:9-46) -- on top of the functions below, and any other
 * language binds the same symbols through its FFI.
 *
 * Conventions
 *   - plain pointers and sizes, no C++ or framework types;
 *   - every function returns 0 (BWTGPU_OK) or a negative BWTGPU_ERR_* code, never
 *     throws or aborts; bwtgpu_last_error() returns a thread-local message;
 *   - there is NO CPU fallback: without a usable HIP device every compute entry
 *     point fails with BWTGPU_ERR_NO_DEVICE;
 *   - block framing is the reference's: a forward block of `len` input bytes
 *     becomes `len + 1` output bytes that contain the delimiter exactly once, at
 *     the row of the suffix that starts at offset 0 (reference src/bwt.cpp:102-123);
 *     blocks are independent, block i of a batch reads `in + i*block_size` and
 *     writes `out + i*(block_size+1)`; only the last block of a batch may be short.
 *     The on-disk container adds one leading delimiter byte per file
 *     (reference src/bwt.cpp:131-139); that byte is written by the host layer, not here.
 *   - "_device" entry points take device pointers, enqueue on `stream` (a hipStream_t
 *     passed as void*, NULL = the default stream) and return without synchronising; the
 *     others take host pointers and are synchronous.  (On a device that cannot launch
 *     cooperative kernels -- no MI300 / MI355 is one -- bwtgpu_forward_batch_device falls back,
 *     for batches of few large blocks, to a variant that reads small results back and blocks.)
 */
#ifndef BWTGPU_H
#define BWTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BWTGPU_OK 0
#define BWTGPU_ERR_NO_DEVICE (-1)      /* no HIP device / bad device index */
#define BWTGPU_ERR_INVALID (-2)        /* bad argument (NULL pointer, block_size == 0, ...) */
#define BWTGPU_ERR_BLOCK_TOO_LARGE (-3)/* block_size above BWTGPU_MAX_BLOCK_SIZE */
#define BWTGPU_ERR_WORKSPACE (-4)      /* workspace too small */
#define BWTGPU_ERR_HIP (-5)            /* a HIP runtime call failed (see bwtgpu_last_error) */
#define BWTGPU_ERR_NO_DELIMITER (-6)   /* inverse: a block does not contain the delimiter */
#define BWTGPU_ERR_NO_FREE_BYTE (-7)   /* all 256 byte values occur: no delimiter can be chosen */
#define BWTGPU_ERR_INTERNAL (-8)       /* a kernel reported an inconsistent state */

/* Largest supported block_size.  Blocks of up to 2 MiB - 1 have dedicated kernels (21-bit suffix
 * indices, three fields per 64-bit sort key); longer ones always take the device-wide forward
 * path and 64-bit inverse table words, and one "_device" call may then hold at most 2^31 bytes. */
#define BWTGPU_MAX_BLOCK_SIZE ((size_t)1 << 30)
/* A batch no longer than one block is one block of its own length, so for such a batch any
 * block_size is accepted (the reference takes any positive value, src/bwt.cpp:251); the limit
 * applies to batches that really are cut into blocks longer than it. */

/* Value stored in the inverse's per-block length array for a block without delimiter. */
#define BWTGPU_LEN_NO_DELIMITER 0xFFFFFFFFu

/* ---- lifecycle --------------------------------------------------------- */

/* Binds the library to the first `n_devices` visible HIP devices (0 = all): per device a staging stream,
 * a pinned status word, launch geometry and the address of the static scratch are set up here, so that
 * no later call has anything left to create on a bound device.  Returns the number of devices in use
 * (> 0) or a negative error.  Idempotent.  Every other entry point binds the devices itself on first
 * use EXCEPT bwtgpu_forward_batch_static, which requires this call to have been made. */
int bwtgpu_init(int n_devices);
/* Waits for the work queued on every bound device, then releases cached device buffers and streams.
 * Call it only after every stream that was handed to a "_device" entry point has been synchronised or
 * destroyed: work still queued may write library-owned memory that this call frees. */
void bwtgpu_shutdown(void);
/* Number of devices bound by bwtgpu_init (0 before init / without a GPU). */
int bwtgpu_device_count(void);
/* Message for the calling thread's most recent error ("" if none). */
const char* bwtgpu_last_error(void);

/* ---- sizes -------------------------------------------------------------- */

/* Blocks in a forward batch: ceil(total_len / block_size). */
size_t bwtgpu_num_blocks(size_t total_len, size_t block_size);
/* Bytes written by a forward batch: total_len + number of blocks. */
size_t bwtgpu_forward_output_bytes(size_t total_len, size_t block_size);
/* Device scratch a forward / inverse "_device" call needs for this batch shape. */
size_t bwtgpu_forward_workspace_bytes(int device, size_t total_len, size_t block_size);
size_t bwtgpu_inverse_workspace_bytes(int device, size_t total_len_with_delims, size_t block_size);

/* ---- forward: replaces the worker pool of reference src/bwt.cpp:143-156,
 *      each worker's bwt_forward (src/bwt.cpp:102-123) and build_suffix_array
 *      (src/bwt.cpp:50-99), for a whole batch of blocks in one launch ---------- */
int bwtgpu_forward_batch_device(int device, const uint8_t* d_in, size_t total_len, size_t block_size,
                                uint8_t delimiter, uint8_t* d_out, void* d_workspace,
                                size_t workspace_bytes, void* stream);
/* Host buffers: `out` receives total_len + nblocks bytes. */
int bwtgpu_forward_batch(int device, const uint8_t* in, size_t total_len, size_t block_size,
                         uint8_t delimiter, uint8_t* out);

/* ---- forward with no allocation anywhere (no counterpart in the reference) -----------------
 * For callers that cannot allocate device memory or be handed a workspace: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  Same transform and same bytes as
 * bwtgpu_forward_batch_device, but the scratch is a fixed array of BWTGPU_STATIC_SCRATCH_BYTES that
 * lives in the library's code object on every GPU (each GPU holds its copy until the process ends),
 * and the call itself allocates nothing, on host or device -- bwtgpu_init must have bound the device
 * before (BWTGPU_ERR_NO_DEVICE otherwise: this call never binds one, which would create a stream and
 * pinned memory).  Limits: block_size (after clamping to
 * the batch) at most 2 MiB - 1 (BWTGPU_ERR_BLOCK_TOO_LARGE), and the batch's block lists (8 bytes per
 * block) plus one block's slice (20 bytes per block byte) must fit the array (BWTGPU_ERR_WORKSPACE).
 * Throughput is lower than with a caller-sized workspace: only as many blocks are in flight as
 * slices fit (49 at 64 KiB blocks, 12 at 256 KiB, 3 at 1 MiB); batches of few large blocks do not
 * take the device-wide path.  All calls on one GPU share the array: calls that overlap on one GPU
 * (different streams or threads) race -- serialise them.  d_status (device memory, may be NULL)
 * receives the kernels' error flags (0 = none) once the stream reaches that point. */
#define BWTGPU_STATIC_SCRATCH_BYTES ((size_t)64 << 20)
int bwtgpu_forward_batch_static(int device, const uint8_t* d_in, size_t total_len, size_t block_size,
                                uint8_t delimiter, uint8_t* d_out, uint32_t* d_status, void* stream);

/* ---- inverse: replaces the serial loop of reference src/inverse_bwt.cpp:85-95 and
 *      bwt_inverse (src/inverse_bwt.cpp:13-63) for a whole batch of blocks ------
 * `in` is the concatenation of (len_i + 1)-byte transformed blocks, i.e. chunks of
 * block_size + 1 bytes, the last possibly shorter.  Block i's text is written so
 * that it ENDS at out + i*block_size + len_i; d_out_lens[i] receives its length,
 * which is len_i for every block produced by the forward transform (then the text
 * starts exactly at out + i*block_size).  A block with more than one delimiter
 * yields a shorter text (the reference stops at the first delimiter it walks into,
 * src/inverse_bwt.cpp:55-57); a block with none yields BWTGPU_LEN_NO_DELIMITER. */
int bwtgpu_inverse_batch_device(int device, const uint8_t* d_in, size_t total_len_with_delims,
                                size_t block_size, uint8_t delimiter, uint8_t* d_out,
                                uint32_t* d_out_lens, void* d_workspace, size_t workspace_bytes,
                                void* stream);
/* Host buffers: `out` must hold total_len_with_delims bytes; *out_len receives the
 * number of bytes produced (texts of all blocks, concatenated and compacted). */
int bwtgpu_inverse_batch(int device, const uint8_t* in, size_t total_len_with_delims,
                         size_t block_size, uint8_t delimiter, uint8_t* out, size_t* out_len);

/* ---- inverse of origin-pointer blocks: the block format of the reference's bzip2 comparison tool
 *      (reference util/bzip2_bwt_utils.cpp:137-198 container, :200-253 inverse).  A block of len text
 *      bytes is the last column of its len sorted ROTATIONS (no delimiter, len bytes) plus the row of
 *      the text itself.  `in` holds the last columns back to back (block i at in + i*block_size, the
 *      last one possibly shorter), orig_ptr[i] the rows; block i's text goes to out + i*block_size.
 *      The container's framing (0xFF, 3-byte big-endian row, last column) is the host layer's
 *      business.  A row outside its block fails with BWTGPU_ERR_INVALID (the reference stops reading
 *      there, :189-192); d_out_lens[i] receives the block's length, or BWTGPU_LEN_NO_DELIMITER for such
 *      a block.  Same kernels as the delimiter format: the given row plays the delimiter row's part. */
size_t bwtgpu_inverse_origin_workspace_bytes(int device, size_t total_len, size_t block_size);
int bwtgpu_inverse_origin_batch_device(int device, const uint8_t* d_in, size_t total_len, size_t block_size,
                                       const uint32_t* d_orig_ptr, uint8_t* d_out, uint32_t* d_out_lens,
                                       void* d_workspace, size_t workspace_bytes, void* stream);
int bwtgpu_inverse_origin_batch(int device, const uint8_t* in, size_t total_len, size_t block_size,
                                const uint32_t* orig_ptr, uint8_t* out);

/* ---- forward of origin-pointer blocks: what the comparison tool's process_block produces (reference
 *      util/bzip2_bwt_utils.cpp:104-136: the block's rotations sorted -- there by BZ2_blockSort, bzip2
 *      1.0.x --, extract_bwt_output :77-96: the byte before each sorted rotation, the block's last byte
 *      for rotation 0, and origPtr = the row of rotation 0).  Block i of `in` (block_size bytes, the last
 *      one possibly shorter) gives its last column at out + i*block_size and its row in rows[i].
 *      Rotations are ordered as unsigned bytes; equal rotations (a block that is a shorter string
 *      repeated) by position, so rows[i] is the first of the tied rows -- any choice among them inverts
 *      to the same text.  The whole batch is sorted by all compute units together (prefix doubling on
 *      ranks read modulo the block length); the call drives the rounds from the host and blocks until
 *      the last one is enqueued.  At most 2^31 bytes per batch; 28 bytes of workspace per input byte. */
size_t bwtgpu_forward_origin_workspace_bytes(int device, size_t total_len, size_t block_size);
int bwtgpu_forward_origin_batch_device(int device, const uint8_t* d_in, size_t total_len, size_t block_size,
                                       uint8_t* d_out, uint32_t* d_rows, void* d_workspace, size_t workspace_bytes,
                                       void* stream);
int bwtgpu_forward_origin_batch(int device, const uint8_t* in, size_t total_len, size_t block_size, uint8_t* out,
                                uint32_t* rows);

/* ---- slab pipeline: replaces the reader -> BlockingQueue -> workers -> ReorderBuffer ->
 *      writer threads of reference src/bwt.cpp:183-233 (and the serial read/transform/write
 *      loop of src/inverse_bwt.cpp:85-95).  The caller fills a pinned input slab with whole
 *      blocks, submits it, and later collects the transformed slab; every slot owns a HIP
 *      stream, so the upload of slab k+1, the kernels of slab k and the download of slab k-1
 *      overlap.  Order is implicit: block i of a slab lands at its closed-form offset, and
 *      the caller collects slots in submission order. -------------------------------------- */
typedef struct bwtgpu_pipeline bwtgpu_pipeline;
#define BWTGPU_FORWARD 0
#define BWTGPU_INVERSE 1
/* `slab_blocks` blocks per slab, `depth` slots in flight (>= 1).  NULL on error. */
bwtgpu_pipeline* bwtgpu_pipeline_create(int device, int direction, size_t block_size, size_t slab_blocks,
                                        int depth);
/* Pinned host buffer of slot `slot`; capacity bwtgpu_pipeline_input_capacity() bytes. */
uint8_t* bwtgpu_pipeline_input(bwtgpu_pipeline* p, int slot);
size_t bwtgpu_pipeline_input_capacity(const bwtgpu_pipeline* p);
/* Enqueues upload + transform + download of the first `bytes` bytes of the slot's input. */
int bwtgpu_pipeline_submit(bwtgpu_pipeline* p, int slot, size_t bytes, uint8_t delimiter);
/* bwtgpu_pipeline_submit and bwtgpu_pipeline_collect in one call, for a caller with nothing to do in between (a
 * file of one slab).  Knowing that the caller waits, the library may pick a route that answers sooner but
 * occupies the calling thread meanwhile (a batch of few large blocks is then sorted by a sequence of launches
 * driven from the host, which beats the resident kernel that bwtgpu_pipeline_submit enqueues). */
int bwtgpu_pipeline_run(bwtgpu_pipeline* p, int slot, size_t bytes, uint8_t delimiter, const uint8_t** out,
                        size_t* out_bytes);
/* Uploads the first `bytes` bytes of the slot's input and returns their byte-presence mask
 * (as bwtgpu_find_unused_byte does; synchronous).  A bwtgpu_pipeline_submit of the same slot
 * with the same byte count that follows uses the uploaded bytes instead of uploading again:
 * a file that fits one slab is read and uploaded once for both the delimiter search
 * (reference src/bwt.cpp:19-47 reads the file a first time for it) and the transform. */
int bwtgpu_pipeline_upload_presence(bwtgpu_pipeline* p, int slot, size_t bytes, uint32_t presence_out[8]);
/* The same for a file of many slabs (forward pipelines): the delimiter needs the whole file's byte values
 * before the first block can be transformed, which the reference settles with a first pass over the file
 * (src/bwt.cpp:19-47).  Here every slab is read and uploaded ONCE and stays on the device until the
 * delimiter is known; with several GPUs each keeps the slabs it will transform.
 *   bwtgpu_pipeline_reserve          device memory for `nslabs` slabs of the pipeline's capacity (kept and
 *                                    reused by later calls that ask for no more); clears the presence mask;
 *   bwtgpu_pipeline_stage            enqueues the upload of the first `bytes` bytes of the slot's input as
 *                                    resident slab `k` and folds their byte values into the mask;
 *   bwtgpu_pipeline_stage_wait       returns once that upload has left the slot's input buffer, which may
 *                                    then be refilled (no-op for a slot with nothing staged);
 *   bwtgpu_pipeline_presence         waits for everything staged and returns the mask;
 *   bwtgpu_pipeline_submit_resident  bwtgpu_pipeline_submit with resident slab `k` as the input; the result
 *                                    is collected from the slot as usual. */
int bwtgpu_pipeline_reserve(bwtgpu_pipeline* p, size_t nslabs);
int bwtgpu_pipeline_stage(bwtgpu_pipeline* p, int slot, size_t k, size_t bytes);
int bwtgpu_pipeline_stage_wait(bwtgpu_pipeline* p, int slot);
int bwtgpu_pipeline_presence(bwtgpu_pipeline* p, uint32_t presence_out[8]);
int bwtgpu_pipeline_submit_resident(bwtgpu_pipeline* p, int slot, size_t k, uint8_t delimiter);
/* Blocks until the slot is done; *out / *out_bytes describe its pinned result, valid until
 * the slot is submitted again.  Inverse results are already compacted. */
int bwtgpu_pipeline_collect(bwtgpu_pipeline* p, int slot, const uint8_t** out, size_t* out_bytes);
void bwtgpu_pipeline_destroy(bwtgpu_pipeline* p);

/* ---- delimiter search: replaces reference src/bwt.cpp:19-47 (find_unique_char) --
 * presence[8] is a 256-bit "byte value occurs" mask (bit v%32 of word v/32).  The
 * device variant ORs into d_presence (zero it first); callers holding several
 * shards OR the masks.  Returns the lowest absent value, or
 * BWTGPU_ERR_NO_FREE_BYTE. */
int bwtgpu_presence_device(int device, const uint8_t* d_data, size_t len, uint32_t* d_presence,
                           void* stream);
int bwtgpu_find_unused_byte(const uint8_t* data, size_t len, uint32_t presence_out[8]);
int bwtgpu_lowest_clear_bit(const uint32_t presence[8]);

/* ---- synthetic inputs (benchmark / test plumbing) ----------------------------
 * Writes bytes [offset, offset+len) of stream (kind, seed) to d_out; kinds:
 * 0 random255, 1 acgt, 2 period7, 3 period1, 4 period64, 5 mixed, 6 text tiled (5 and 6 need d_text). */
int bwtgpu_synth_device(int device, int kind, uint64_t seed, uint64_t offset, uint8_t* d_out,
                        size_t len, const uint8_t* d_text, size_t text_len, void* stream);

/* ---- measurement plumbing (reference tests/performance.cpp:84-117 times file to file only) ----
 * Times the transform with the data resident in device memory: uploads `in` (at most 1 GiB) once, runs
 * one untimed and `reps` timed forward passes, then the same for the inverse over the forward result,
 * bracketed by HIP events on the device's stream.  *forward_ms / *inverse_ms receive the mean time of
 * one pass in milliseconds.  Used by the performance tool for its device-resident columns. */
int bwtgpu_time_resident(int device, const uint8_t* in, size_t total_len, size_t block_size, uint8_t delimiter,
                         int reps, double* forward_ms, double* inverse_ms);

/* ---- pinned host memory for callers that stage their own batches --------------- */
void* bwtgpu_alloc_host(size_t bytes);
void bwtgpu_free_host(void* p);

#ifdef __cplusplus
}
#endif
#endif /* BWTGPU_H */
