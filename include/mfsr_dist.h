/*
 * mfsr_dist.h -- multi-GPU bursts: one process (or thread) per GPU of one node, RCCL over xGMI.
 *
 * The reference is single-GPU (cudaSetDevice(0), test_opencv/kernel.cu:45; no NCCL/MPI call site anywhere), so this
 * boundary has no reference counterpart: it is the C-ABI of the frame-sharded burst that BASELINE.json's north_star
 * asks for ("partition alignment+warp over the 8 GPUs of one node with an RCCL reduce onto rank 0's accumulation
 * buffers over xGMI").  libmfsr_dist.so links librccl and libmfsr_hip.so; everything below is built on the burst
 * building blocks of include/mfsr.h (mfsr_burst_align_frame / mfsr_burst_fuse_rows / mfsr_burst_finish_rows).
 *
 * Sharding: rank g ALIGNS frames {k : k mod world == g} (stages A1, I, B, D, F); the reference-frame products are
 * LR-sized and are computed redundantly by every rank.  Three ways to combine:
 *
 *   MFSR_DIST_STRIPES (default)  fuse is sharded over HR row stripes: each rank sends every other rank the rows of
 *       its frames' raw / flow / certainty (.xyz: the fuse never reads .w) that the peer's stripe reads (point-to-point
 *       ncclSend/ncclRecv, one per xGMI link), then fuses ALL frames in frame order onto its own stripe, normalises it, and rank 0 collects the
 *       u16 stripes.  Traffic per rank ~ N * LR * 38 B / world (0.13 GB at 4K x4, 16 frames, 8 GPUs) instead of the
 *       2.8 GB of accumulators a reduce(-scatter) moves, and the summation order is the single-GPU one: the result is
 *       BIT-IDENTICAL to the 1-GPU burst.
 *   MFSR_DIST_REDUCE             every rank fuses its own frames onto private full-size accumulators, ncclReduce(sum)
 *       of both onto rank 0, rank 0 finishes: north_star's literal wording.  Root-bound.
 *   MFSR_DIST_REDUCE_SCATTER     as REDUCE with ncclReduceScatter over HR row stripes, stripe finish, u16 gather.
 *   (REDUCE / REDUCE_SCATTER add the per-rank partial sums in another order than one GPU does: equal to fp32 rounding.)
 *
 * Conventions as in mfsr.h: int status (0 ok, >0 hipError_t, <0 MFSR_E_*; RCCL failures map to MFSR_E_COMM), nothing
 * throws, all work is enqueued on the caller's stream, the caller owns every buffer.
 */
#ifndef MFSR_DIST_H
#define MFSR_DIST_H

#include "mfsr.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { MFSR_DIST_STRIPES = 0, MFSR_DIST_REDUCE = 1, MFSR_DIST_REDUCE_SCATTER = 2 };
#define MFSR_E_COMM (-5)        /* a transport call failed: RCCL error, or a peer of a local group failed / timed out (logged) */
#define MFSR_DIST_ID_BYTES 128  /* sizeof(ncclUniqueId) */
#define MFSR_DIST_DEFAULT_RAW_HALO 64

typedef struct mfsr_dist mfsr_dist;

/* rank 0 creates the id (ncclGetUniqueId) and hands it to the other ranks by whatever means the launcher has */
int mfsr_dist_get_unique_id(void* id);
/* device bytes one rank needs for cfg (burst workspace, accumulators, per-frame products of all cfg->frames frames,
 * u16 staging image) */
size_t mfsr_dist_workspace_bytes(const mfsr_config* cfg, int worldSize);
/* ncclCommInitRank on the CURRENT device + burst context.  cfg->frames = frames of the whole burst. */
int mfsr_dist_create(mfsr_dist** out, const mfsr_config* cfg, int rank, int worldSize, const void* id, void* workspace,
                     size_t workspaceBytes);
void mfsr_dist_destroy(mfsr_dist* d);
/* the (first) burst context a mfsr_dist drives (mfsr_burst_debug_views); owned by d.  Pipelined bursts (below) alternate
 * between two contexts: time the warp+fuse launches of a rank with mfsr_dist_timing / mfsr_dist_timing_read, which cover both
 * (same meaning as mfsr_burst_timing / mfsr_burst_timing_read). */
mfsr_burst* mfsr_dist_burst(mfsr_dist* d);
int mfsr_dist_timing(mfsr_dist* d, int enable);
int mfsr_dist_timing_read(mfsr_dist* d, double* totalMs, int* launches, int* frames);
/* raw-row halo of the STRIPES exchange (default 64: vertical flow up to 61 raw pixels).  A halo >= the frame height
 * exchanges whole raw frames: always valid, ~2.5x the traffic (still ~9x below summing accumulators) -- the fallback
 * when mfsr_dist_process_burst reports status 1. */
int mfsr_dist_set_raw_halo(mfsr_dist* d, int rawHalo);
/* HR rows of `rank` in the STRIPES / REDUCE_SCATTER modes */
int mfsr_dist_stripe(const mfsr_dist* d, int rank, int* rowBegin, int* rowEnd);
/* One burst.  frames[k] = device pointer of frame k (dense u16) for the frames this rank owns (k mod world == rank) and
 * for the reference frame on EVERY rank; other entries are ignored.  out16: dense interleaved u16 HR image, written on
 * rank 0 only (may be NULL elsewhere).  status: device int, set to 0 / 1 on every rank: 1 = a frame's vertical flow
 * exceeded the raw halo of the STRIPES exchange (result invalid: raise the halo or use another mode).
 *
 * WHEN THE OUTPUT EXISTS.  The call is asynchronous and, like mfsr_dist_wait_output, a collective: every rank makes it, in
 * the same order.
 *   - frames[]: read by work enqueued on `stream` only -- the buffers may be refilled once `stream` has passed this call,
 *     as with any asynchronous call (pipelined bursts keep their own copy of the rank's frames).
 *   - REDUCE / REDUCE_SCATTER, a world of one rank, or MFSR_DIST_PIPELINE=0: out16 / status are complete once `stream`
 *     has passed mfsr_dist_wait_output.
 *   - STRIPES with world > 1 (the default) runs bursts PIPELINED: call i enqueues the reference products, the alignment
 *     and the exchange of burst i, then the fuse / finish / gather of burst i - 1, whose exchange finished while burst i
 *     was being aligned (the caller's stream never waits for an exchange).  out16 / status of call i are therefore written
 *     by the work that call i + 1 -- or mfsr_dist_wait_output -- enqueues: the POINTERS must stay valid and untouched until
 *     then, and the image of burst i is complete only once `stream` has passed mfsr_dist_wait_output (last burst) or
 *     mfsr_dist_wait_previous (burst i after call i + 1).  A device-wide synchronise after call i alone does NOT
 *     produce image i.
 * The transport calls go to a stream the context owns (same order on every rank), event-linked to `stream`.
 * MFSR_DIST_OVERLAP=0 in the environment keeps every call on the caller's stream (and turns the pipelining off). */
int mfsr_dist_process_burst(mfsr_dist* d, const uint16_t* const* frames, int mode, uint16_t* out16, int* status,
                            mfsr_stream_t stream);
/* enqueues what is still owed of the last burst (pipelined: its fuse / finish / gather -- transport calls, so EVERY rank
 * calls this, like process_burst) and makes `stream` wait for its output */
int mfsr_dist_wait_output(mfsr_dist* d, mfsr_stream_t stream);
/* makes `stream` wait for the newest output whose work HAS been enqueued (pipelined: burst i - 1 after call i); no
 * transport call, not a collective, the pipeline keeps running */
int mfsr_dist_wait_previous(mfsr_dist* d, mfsr_stream_t stream);
/* largest |vertical flow| (raw pixels) over all frames of the newest burst whose back half has been enqueued, measured on
 * the device and max-reduced over the ranks; waits for that burst's gather and synchronises `stream`.  A caller sizes the
 * raw-row halo of the following bursts from it: mfsr_dist_set_raw_halo(d, ceil(v) + 3 + margin) instead of the default 64
 * rows -- status 1 still guards every burst. */
int mfsr_dist_measured_flow(mfsr_dist* d, float* maxAbsFlowY, mfsr_stream_t stream);
/* "rccl" or "local": the transport behind d (see below) */
const char* mfsr_dist_transport(const mfsr_dist* d);
/* messages and bytes this rank SENT during the last mfsr_dist_process_burst (exchange + stripe gather; the collectives of
 * the reduce modes and the one-int status all-reduce are not counted).  STRIPES: one packed message per peer whose stripe
 * is not empty (+ one stripe to rank 0), whatever the number of frames. */
int mfsr_dist_exchange_stats(const mfsr_dist* d, long long* messagesSent, long long* bytesSent);

/* ---- G ranks inside ONE process ("local" transport) -------------------------------------------------------------------
 * The reference's entry point is one process with one thread (finalProject/Project/multi_frame_sr.cpp:122-210); a drop-in
 * of it cannot start one process per GPU.  mfsr_dist_group runs the SAME per-rank code as mfsr_dist_create's contexts
 * (csrc/dist.cpp: process_stripes / process_reduce / gather_stripes behind one transport table) with one worker thread
 * per rank and peer copies (hipMemcpyPeerAsync, ordered by events after a host-side rendezvous) in place of RCCL calls.
 * devices[r] = HIP device of rank r (NULL: every rank on the current device).  Several ranks may share a device
 * ("virtual ranks"): that is how the multi-rank code is exercised on a one-GPU box.  workspaces[r]: workspaceBytes >=
 * mfsr_dist_workspace_bytes(cfg, worldSize) device bytes on devices[r], 256-byte aligned. */
typedef struct mfsr_dist_group mfsr_dist_group;
int mfsr_dist_group_create(mfsr_dist_group** out, const mfsr_config* cfg, int worldSize, const int* devices, void* const* workspaces,
                           size_t workspaceBytes);
void mfsr_dist_group_destroy(mfsr_dist_group* g);
mfsr_dist* mfsr_dist_group_rank(mfsr_dist_group* g, int rank); /* owned by g */
int mfsr_dist_group_set_raw_halo(mfsr_dist_group* g, int rawHalo);
/* One burst on all ranks.  frames: worldSize x cfg->frames table (row r = the `frames` argument of mfsr_dist_process_burst
 * for rank r: pointers on devices[r]); out16 on devices[0]; status: per-rank device ints (or NULL); streams: per-rank
 * streams (or NULL: streams the group owns -- ranks sharing a device must NOT share a stream, and never the NULL stream).
 * Returns when every rank has ENQUEUED its burst (asynchronous like mfsr_dist_process_burst). */
int mfsr_dist_group_process_burst(mfsr_dist_group* g, const uint16_t* const* frames, int mode, uint16_t* out16, int* const* status,
                                  const mfsr_stream_t* streams);
/* every rank's stream waits for the last burst's output (mfsr_dist_wait_output per rank) */
int mfsr_dist_group_wait_output(mfsr_dist_group* g, const mfsr_stream_t* streams);
/* blocks the host until every rank's streams (the given ones, or the group's own) and comm streams are idle */
int mfsr_dist_group_synchronize(mfsr_dist_group* g, const mfsr_stream_t* streams);

#ifdef __cplusplus
}
#endif
#endif /* MFSR_DIST_H */
