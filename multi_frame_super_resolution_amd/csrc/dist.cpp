// dist.cpp -- multi-GPU bursts over RCCL (include/mfsr_dist.h): frame-sharded alignment, then either HR-stripe
// sharded fuse after a point-to-point exchange of the LR-sized per-frame products (default), or private accumulators
// summed with ncclReduce / ncclReduceScatter.  One mfsr_dist per GPU; all work goes to the caller's stream.
//
// The reference has no multi-GPU code at all (cudaSetDevice(0), test_opencv/kernel.cu:45): nothing here follows a
// reference call pattern.  xGMI is point-to-point (7 links per GPU), which is what the STRIPES exchange is shaped
// for: every rank talks to every peer at once with messages of a few MB, one stream of traffic per link, instead of
// funnelling 2 x HR x 12 B per rank through a ring or into one root.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mfsr_dist.h"

static_assert(sizeof(ncclUniqueId) == MFSR_DIST_ID_BYTES, "MFSR_DIST_ID_BYTES must be sizeof(ncclUniqueId)");

#define D_REQUIRE(cond)                                                                         \
    do {                                                                                        \
        if (!(cond)) {                                                                          \
            fprintf(stderr, "mfsr_dist: invalid argument: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
            return MFSR_E_INVALID;                                                              \
        }                                                                                       \
    } while (0)
#define D_TRY(expr)                       \
    do {                                  \
        int rc_ = (expr);                 \
        if (rc_ != MFSR_OK) return rc_;   \
    } while (0)
#define D_HIP(expr)                                                                                          \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            fprintf(stderr, "mfsr_dist: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                                                  \
        }                                                                                                    \
    } while (0)
#define D_NCCL(expr)                                                                                           \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) {                                                                               \
            fprintf(stderr, "mfsr_dist: %s failed: %s (%s:%d)\n", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
            return MFSR_E_COMM;                                                                                \
        }                                                                                                      \
    } while (0)

namespace {

inline size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct DistLayout {
    int W, H, s, N, hrW, hrH, tw, th, hw, hh;
    int flowPitch, maskPitch;
    size_t burstWs, accBytes, rawBytes, flowBytes, maskBytes, out16Bytes;
    size_t offBurst, offImg, offTw, offRaw, offFlow, offMask, offOut16, offFlag, total;
};

int make_layout(const mfsr_config* c, DistLayout* L)
{
    memset(L, 0, sizeof(*L));
    L->W = c->width;
    L->H = c->height;
    L->s = c->scale;
    L->N = c->frames;
    L->hrW = c->width * c->scale;
    L->hrH = c->height * c->scale;
    L->hw = c->width / 2;
    L->hh = c->height / 2;
    L->tw = c->mono ? c->width : L->hw;
    L->th = c->mono ? c->height : L->hh;
    L->flowPitch = L->tw * 8;    // dense rows: a row range is one contiguous message
    L->maskPitch = L->hw * 16;
    L->burstWs = mfsr_burst_workspace_bytes(c);
    if (L->burstWs == 0) return MFSR_E_INVALID;
    L->accBytes = mfsr_burst_accumulator_bytes(c);
    L->rawBytes = (size_t)L->W * L->H * 2;
    L->flowBytes = (size_t)L->flowPitch * L->th;
    L->maskBytes = (size_t)L->maskPitch * L->hh;
    L->out16Bytes = (size_t)L->hrW * L->hrH * 6;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        off = up(off, 256);
        const size_t o = off;
        off += bytes;
        return o;
    };
    L->offBurst = take(L->burstWs);
    L->offImg = take(L->accBytes);
    L->offTw = take(L->accBytes);
    L->offRaw = take(up(L->rawBytes, 256) * L->N);
    L->offFlow = take(up(L->flowBytes, 256) * L->N);
    L->offMask = take(up(L->maskBytes, 256) * L->N);
    L->offOut16 = take(L->out16Bytes);
    L->offFlag = take(256);
    L->total = up(off, 256);
    return MFSR_OK;
}

}  // namespace

struct mfsr_dist {
    mfsr_config cfg;
    DistLayout L;
    int rank, world, rawHalo;
    ncclComm_t comm;
    mfsr_burst* burst;
    char* base;
    mfsr_float3 *imgOut, *totalWeights;
    uint16_t* out16;
    int* flag;   // two ints: bursts alternate (the previous burst's all-reduce may still be in flight on the comm stream)
    // STRIPES: every RCCL call goes to commStream, in the same order on every rank (exchange i, gather i, exchange i+1, ..),
    // event-linked to the caller's stream, so that the gather of burst i overlaps the alignment of burst i+1
    hipStream_t commStream;
    hipEvent_t evAligned, evExchanged, evFinished, evGathered;
    bool gatherPending;
    int overlap;
    long long burstNo;
    uint16_t* raw(int k) const { return (uint16_t*)(base + L.offRaw + up(L.rawBytes, 256) * (size_t)k); }
    mfsr_float2* flow(int k) const { return (mfsr_float2*)(base + L.offFlow + up(L.flowBytes, 256) * (size_t)k); }
    mfsr_float4* mask(int k) const { return (mfsr_float4*)(base + L.offMask + up(L.maskBytes, 256) * (size_t)k); }
};

extern "C" int mfsr_dist_get_unique_id(void* id)
{
    D_REQUIRE(id != nullptr);
    ncclUniqueId u;
    D_NCCL(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return MFSR_OK;
}

extern "C" size_t mfsr_dist_workspace_bytes(const mfsr_config* cfg, int worldSize)
{
    DistLayout L;
    if (!cfg || worldSize < 1 || make_layout(cfg, &L) != MFSR_OK) return 0;
    return L.total;
}

extern "C" int mfsr_dist_create(mfsr_dist** out, const mfsr_config* cfg, int rank, int worldSize, const void* id, void* workspace,
                                size_t workspaceBytes)
{
    D_REQUIRE(out && cfg && id && workspace && worldSize >= 1 && rank >= 0 && rank < worldSize);
    D_REQUIRE(((uintptr_t)workspace & 255) == 0);
    mfsr_dist* d = new (std::nothrow) mfsr_dist;
    D_REQUIRE(d != nullptr);
    memset((void*)d, 0, sizeof(*d));
    d->cfg = *cfg;
    int rc = make_layout(cfg, &d->L);
    if (rc == MFSR_OK && d->L.total > workspaceBytes) {
        fprintf(stderr, "mfsr_dist: workspace too small: need %zu bytes, got %zu\n", d->L.total, workspaceBytes);
        rc = MFSR_E_WORKSPACE;
    }
    if (rc != MFSR_OK) {
        delete d;
        return rc;
    }
    d->rank = rank;
    d->world = worldSize;
    d->rawHalo = MFSR_DIST_DEFAULT_RAW_HALO;
    d->base = (char*)workspace;
    d->imgOut = (mfsr_float3*)(d->base + d->L.offImg);
    d->totalWeights = (mfsr_float3*)(d->base + d->L.offTw);
    d->out16 = (uint16_t*)(d->base + d->L.offOut16);
    d->flag = (int*)(d->base + d->L.offFlag);
    rc = mfsr_burst_create(&d->burst, cfg, d->base + d->L.offBurst, d->L.burstWs);
    if (rc != MFSR_OK) {
        delete d;
        return rc;
    }
    {
        const char* e = getenv("MFSR_DIST_OVERLAP");
        d->overlap = (e && e[0] == '0') ? 0 : 1;
        hipError_t he = hipStreamCreateWithFlags(&d->commStream, hipStreamNonBlocking);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&d->evAligned, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&d->evExchanged, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&d->evFinished, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&d->evGathered, hipEventDisableTiming);
        if (he != hipSuccess) {
            fprintf(stderr, "mfsr_dist: stream / event creation failed: %s\n", hipGetErrorString(he));
            mfsr_burst_destroy(d->burst);
            delete d;
            return (int)he;
        }
    }
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t r = ncclCommInitRank(&d->comm, worldSize, u, rank);
    if (r != ncclSuccess) {
        fprintf(stderr, "mfsr_dist: ncclCommInitRank failed: %s\n", ncclGetErrorString(r));
        mfsr_burst_destroy(d->burst);
        delete d;
        return MFSR_E_COMM;
    }
    *out = d;
    return MFSR_OK;
}

extern "C" void mfsr_dist_destroy(mfsr_dist* d)
{
    if (!d) return;
    if (d->commStream) (void)hipStreamSynchronize(d->commStream);
    if (d->comm) (void)ncclCommDestroy(d->comm);
    if (d->evAligned) (void)hipEventDestroy(d->evAligned);
    if (d->evExchanged) (void)hipEventDestroy(d->evExchanged);
    if (d->evFinished) (void)hipEventDestroy(d->evFinished);
    if (d->evGathered) (void)hipEventDestroy(d->evGathered);
    if (d->commStream) (void)hipStreamDestroy(d->commStream);
    mfsr_burst_destroy(d->burst);
    delete d;
}

extern "C" mfsr_burst* mfsr_dist_burst(mfsr_dist* d) { return d ? d->burst : nullptr; }

extern "C" int mfsr_dist_set_raw_halo(mfsr_dist* d, int rawHalo)
{
    D_REQUIRE(d && rawHalo >= 4 && rawHalo <= 4096);
    d->rawHalo = rawHalo;
    return MFSR_OK;
}

extern "C" int mfsr_dist_stripe(const mfsr_dist* d, int rank, int* rowBegin, int* rowEnd)
{
    D_REQUIRE(d && rank >= 0 && rank < d->world);
    mfsr_stripe_plan p;
    D_TRY(mfsr_dist_stripe_plan(&d->cfg, d->world, rank, d->rawHalo, &p));
    if (rowBegin) *rowBegin = p.rowBegin;
    if (rowEnd) *rowEnd = p.rowEnd;
    return MFSR_OK;
}

// rank 0 collects the u16 stripes (stripe p = HR rows [b_p, e_p) of rank p's staging image)
static int gather_stripes(mfsr_dist* d, const std::vector<mfsr_stripe_plan>& plan, uint16_t* out16, hipStream_t st)
{
    const size_t rowBytes = (size_t)d->L.hrW * 6;
    D_NCCL(ncclGroupStart());
    if (d->rank == 0) {
        for (int p = 1; p < d->world; p++) {
            const size_t n = (size_t)(plan[p].rowEnd - plan[p].rowBegin) * rowBytes;
            if (n) D_NCCL(ncclRecv((char*)out16 + (size_t)plan[p].rowBegin * rowBytes, n, ncclUint8, p, d->comm, st));
        }
    } else {
        const size_t n = (size_t)(plan[d->rank].rowEnd - plan[d->rank].rowBegin) * rowBytes;
        if (n) D_NCCL(ncclSend((const char*)d->out16 + (size_t)plan[d->rank].rowBegin * rowBytes, n, ncclUint8, 0, d->comm, st));
    }
    D_NCCL(ncclGroupEnd());
    return MFSR_OK;
}

static int process_stripes(mfsr_dist* d, const uint16_t* const* frames, uint16_t* out16, int* status, hipStream_t st)
{
    const mfsr_config& c = d->cfg;
    const DistLayout& L = d->L;
    const int N = c.frames, G = d->world, me = d->rank, ref = c.reference;
    std::vector<mfsr_stripe_plan> plan(G);
    for (int p = 0; p < G; p++) D_TRY(mfsr_dist_stripe_plan(&c, G, p, d->rawHalo, &plan[p]));
    const mfsr_stripe_plan& mine = plan[me];
    int* flag = d->flag + (d->burstNo++ & 1);
    // A = the caller's stream (kernels), B = the comm stream (every RCCL call); with overlap off B = A
    hipStream_t B = d->overlap ? d->commStream : st;
    D_HIP(hipMemsetAsync(flag, 0, sizeof(int), st));

    // reference products on every rank, then this rank's frames: alignment only
    D_TRY(mfsr_burst_set_reference(d->burst, frames[ref], (mfsr_stream_t)st));
    std::vector<const uint16_t*> raws(N, nullptr);
    for (int k = 0; k < N; k++) {
        if (k % G != me) continue;
        D_REQUIRE(frames[k] != nullptr);
        raws[k] = frames[k];
        D_TRY(mfsr_burst_align_frame(d->burst, frames[k], k == ref, d->flow(k), L.flowPitch, d->mask(k), L.maskPitch, (mfsr_stream_t)st));
    }

    // exchange (on B, after the alignment on A): to peer p the rows of my frames that p's stripe reads; from the owner of
    // frame k the rows mine reads.  (Receive buffers: B is past gather(i-1), which waited for fuse(i-1) -- their last reader.)
    D_HIP(hipEventRecord(d->evAligned, st));
    D_HIP(hipStreamWaitEvent(B, d->evAligned, 0));
    for (int k = 0; k < N; k++)
        if (k % G != me) raws[k] = d->raw(k);
    if (G > 1) {
        hipStream_t st = B;  // the sends / receives below go to the comm stream
        D_NCCL(ncclGroupStart());
        for (int p = 0; p < G; p++) {
            if (p == me || plan[p].rowEnd <= plan[p].rowBegin) continue;
            for (int k = me; k < N; k += G) {
                D_NCCL(ncclSend((const char*)raws[k] + (size_t)plan[p].rawRow0 * L.W * 2, (size_t)plan[p].rawRows * L.W * 2, ncclUint8, p,
                                d->comm, st));
                D_NCCL(ncclSend((const char*)d->flow(k) + (size_t)plan[p].flowRow0 * L.flowPitch, (size_t)plan[p].flowRows * L.flowPitch,
                                ncclUint8, p, d->comm, st));
                D_NCCL(ncclSend((const char*)d->mask(k) + (size_t)plan[p].maskRow0 * L.maskPitch, (size_t)plan[p].maskRows * L.maskPitch,
                                ncclUint8, p, d->comm, st));
            }
        }
        if (mine.rowEnd > mine.rowBegin) {
            for (int k = 0; k < N; k++) {
                const int owner = k % G;
                if (owner == me) continue;
                raws[k] = d->raw(k);
                D_NCCL(ncclRecv((char*)d->raw(k) + (size_t)mine.rawRow0 * L.W * 2, (size_t)mine.rawRows * L.W * 2, ncclUint8, owner, d->comm,
                                st));
                D_NCCL(ncclRecv((char*)d->flow(k) + (size_t)mine.flowRow0 * L.flowPitch, (size_t)mine.flowRows * L.flowPitch, ncclUint8,
                                owner, d->comm, st));
                D_NCCL(ncclRecv((char*)d->mask(k) + (size_t)mine.maskRow0 * L.maskPitch, (size_t)mine.maskRows * L.maskPitch, ncclUint8,
                                owner, d->comm, st));
            }
        }
        D_NCCL(ncclGroupEnd());
    }
    D_HIP(hipEventRecord(d->evExchanged, B));
    D_HIP(hipStreamWaitEvent(st, d->evExchanged, 0));

    // every frame, in frame order, a group per pass over the accumulators, onto this rank's HR rows only
    if (mine.rowEnd > mine.rowBegin) {
        // (a halo that spans the whole frame needs no check: every raw row is present)
        for (int k = 0; k < N && mine.rawRows < c.height; k++)
            D_TRY(mfsr_checkFlowBound((const mfsr_float2*)((const char*)d->flow(k) + (size_t)mine.flowRow0 * L.flowPitch), L.flowPitch, L.tw,
                                      mine.flowRows, mine.maxFlowY, flag, (mfsr_stream_t)st));
        const int per = mfsr_burst_group_size(&c);  // the grouping of the single-GPU burst: same sums in the same order
        for (int k = 0; k < N; k += per) {
            const int n = (k + per <= N) ? per : N - k;
            const uint16_t* rg[MFSR_MAX_FUSE_GROUP];
            const mfsr_float2* fg[MFSR_MAX_FUSE_GROUP];
            const mfsr_float4* mg[MFSR_MAX_FUSE_GROUP];
            for (int j = 0; j < n; j++) {
                rg[j] = raws[k + j];
                fg[j] = d->flow(k + j);
                mg[j] = d->mask(k + j);
            }
            D_TRY(mfsr_burst_fuse_rows(d->burst, n, rg, fg, L.flowPitch, mg, L.maskPitch, d->imgOut, d->totalWeights, k == 0 ? 1 : 0,
                                       mine.rowBegin, mine.rowEnd, (mfsr_stream_t)st));
        }
        uint16_t* dst = me == 0 ? out16 : d->out16;
        // the staging image may still be on its way to rank 0 (gather of the previous burst, on B)
        if (d->gatherPending) D_HIP(hipStreamWaitEvent(st, d->evGathered, 0));
        D_TRY(mfsr_burst_finish_rows(d->burst, d->imgOut, d->totalWeights, nullptr, dst, mine.rowBegin, mine.rowEnd - mine.rowBegin,
                                     (mfsr_stream_t)st));
    }
    // gather + status on B after the finish on A; the caller's stream does NOT wait for it (mfsr_dist_wait_output does):
    // the next burst's alignment runs meanwhile
    D_HIP(hipEventRecord(d->evFinished, st));
    D_HIP(hipStreamWaitEvent(B, d->evFinished, 0));
    if (G > 1) {
        D_TRY(gather_stripes(d, plan, out16, B));
        D_NCCL(ncclAllReduce(flag, flag, 1, ncclInt32, ncclMax, d->comm, B));
    }
    if (status) D_HIP(hipMemcpyAsync(status, flag, sizeof(int), hipMemcpyDeviceToDevice, B));
    D_HIP(hipEventRecord(d->evGathered, B));
    d->gatherPending = true;
    return MFSR_OK;
}

static int process_reduce(mfsr_dist* d, const uint16_t* const* frames, int mode, uint16_t* out16, int* status, hipStream_t st)
{
    const mfsr_config& c = d->cfg;
    const DistLayout& L = d->L;
    const int N = c.frames, G = d->world, me = d->rank, ref = c.reference;
    D_TRY(mfsr_burst_begin(d->burst, d->imgOut, d->totalWeights, (mfsr_stream_t)st));
    D_TRY(mfsr_burst_set_reference(d->burst, frames[ref], (mfsr_stream_t)st));
    for (int k = me; k < N; k += G) {
        D_REQUIRE(frames[k] != nullptr);
        D_TRY(mfsr_burst_add_frame(d->burst, frames[k], k == ref, d->imgOut, d->totalWeights, (mfsr_stream_t)st));
    }
    D_TRY(mfsr_burst_flush(d->burst, (mfsr_stream_t)st));  // the pending frame of an odd shard; zeroes if the shard is empty
    const size_t count = (size_t)L.hrW * L.hrH * 3;
    if (status) D_HIP(hipMemsetAsync(status, 0, sizeof(int), st));
    if (G == 1) return mfsr_burst_finish(d->burst, d->imgOut, d->totalWeights, nullptr, out16, (mfsr_stream_t)st);
    if (mode == MFSR_DIST_REDUCE_SCATTER && (L.hrH % G) == 0) {
        const size_t chunk = count / G;
        const int rows = L.hrH / G;
        float* a = (float*)d->imgOut;
        float* w = (float*)d->totalWeights;
        D_NCCL(ncclGroupStart());
        D_NCCL(ncclReduceScatter(a, a + chunk * me, chunk, ncclFloat, ncclSum, d->comm, st));
        D_NCCL(ncclReduceScatter(w, w + chunk * me, chunk, ncclFloat, ncclSum, d->comm, st));
        D_NCCL(ncclGroupEnd());
        uint16_t* dst = me == 0 ? out16 : d->out16;
        D_TRY(mfsr_burst_finish_rows(d->burst, d->imgOut, d->totalWeights, nullptr, dst, rows * me, rows, (mfsr_stream_t)st));
        std::vector<mfsr_stripe_plan> plan(G);
        for (int p = 0; p < G; p++) {
            memset(&plan[p], 0, sizeof(plan[p]));
            plan[p].rowBegin = rows * p;
            plan[p].rowEnd = rows * (p + 1);
        }
        return gather_stripes(d, plan, out16, st);
    }
    // MFSR_DIST_REDUCE (and REDUCE_SCATTER when the HR rows do not divide by the world size)
    D_NCCL(ncclGroupStart());
    D_NCCL(ncclReduce(d->imgOut, d->imgOut, count, ncclFloat, ncclSum, 0, d->comm, st));
    D_NCCL(ncclReduce(d->totalWeights, d->totalWeights, count, ncclFloat, ncclSum, 0, d->comm, st));
    D_NCCL(ncclGroupEnd());
    if (me == 0) return mfsr_burst_finish(d->burst, d->imgOut, d->totalWeights, nullptr, out16, (mfsr_stream_t)st);
    return MFSR_OK;
}

extern "C" int mfsr_dist_process_burst(mfsr_dist* d, const uint16_t* const* frames, int mode, uint16_t* out16, int* status,
                                       mfsr_stream_t stream)
{
    D_REQUIRE(d && frames);
    D_REQUIRE(mode == MFSR_DIST_STRIPES || mode == MFSR_DIST_REDUCE || mode == MFSR_DIST_REDUCE_SCATTER);
    D_REQUIRE(frames[d->cfg.reference] != nullptr);
    D_REQUIRE(d->rank != 0 || out16 != nullptr);
    hipStream_t st = (hipStream_t)stream;
    if (mode == MFSR_DIST_STRIPES) return process_stripes(d, frames, out16, status, st);
    if (d->gatherPending) {  // a STRIPES burst before: its gather uses the communicator on the comm stream
        D_HIP(hipStreamWaitEvent(st, d->evGathered, 0));
        d->gatherPending = false;
    }
    return process_reduce(d, frames, mode, out16, status, st);
}

extern "C" int mfsr_dist_wait_output(mfsr_dist* d, mfsr_stream_t stream)
{
    D_REQUIRE(d != nullptr);
    if (d->gatherPending) D_HIP(hipStreamWaitEvent((hipStream_t)stream, d->evGathered, 0));
    return MFSR_OK;
}
