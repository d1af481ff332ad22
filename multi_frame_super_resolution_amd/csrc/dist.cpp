// dist.cpp -- multi-GPU bursts (include/mfsr_dist.h): frame-sharded alignment, then either HR-stripe sharded fuse after
// a point-to-point exchange of the LR-sized per-frame products (default), or private accumulators summed with a
// reduce / reduce-scatter.  One mfsr_dist per GPU; all work goes to the caller's stream (+ a comm stream the context owns).
//
// The reference has no multi-GPU code at all (cudaSetDevice(0), test_opencv/kernel.cu:45): nothing here follows a
// reference call pattern.  xGMI is point-to-point (7 links per GPU), which is what the STRIPES exchange is shaped
// for: every rank talks to every peer at once with ONE packed message of a few MB per peer, one stream of traffic per
// link, instead of funnelling 2 x HR x 12 B per rank through a ring or into one root.
//
// Every communication call goes through a small transport table (struct Transport: group / send / recv / all-reduce /
// reduce / reduce-scatter) with two back ends:
//   * RcclTransport  -- one process per GPU, ncclSend/ncclRecv/ncclReduce/... on an ncclComm_t (mfsr_dist_create);
//   * LocalTransport -- G ranks in ONE process (mfsr_dist_group_*): one host thread per rank, messages are peer copies
//                       (hipMemcpyPeerAsync) ordered by events after a host-side rendezvous.  The ranks may sit on G
//                       different GPUs (a single-process multi-GPU burst, apps/multi_frame_sr) or share one device
//                       ("virtual ranks": how the G > 1 code of this file runs under test on a one-GPU box).
// process_stripes / process_reduce / gather_stripes below are the same code for both.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
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
// inside a transport group nothing returns early: the first error is kept and the group is always closed
#define D_KEEP(rc, expr)                      \
    do {                                      \
        int k_ = (expr);                      \
        if ((rc) == MFSR_OK) (rc) = k_;       \
    } while (0)

namespace {

inline size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------------------------------
// small device helpers of this layer (dist.cpp is compiled as HIP)

// kind 0: plain copy of `bytes` bytes; kind 1: `bytes` = bytes of the float4 SOURCE texels, of which .xyz are stored densely
// (12 B per texel); kind 2: the reverse, `bytes` = bytes of the float4 DESTINATION texels, .w := 0.  (The certainty mask's
// .w holds the robustness kernel's M, RobustnessModell.cu:155, which no fuse kernel reads: a quarter of the mask bytes stay
// at home.)
enum { SEG_COPY = 0, SEG_PACK_XYZ = 1, SEG_UNPACK_XYZ = 2 };
struct CopySeg {
    const void* src;
    void* dst;
    unsigned long long bytes;  // multiple of 4 (kind 1, 2: of 16)
    int kind = SEG_COPY;
};
constexpr int kSegsPerLaunch = 96;  // 96 x 32 B of kernel arguments
constexpr int kBlocksPerSeg = 32;
struct CopySegs {
    CopySeg s[kSegsPerLaunch];
};

// blockIdx.y = segment, kBlocksPerSeg workgroups stride over it in 16-byte units (4-byte units if either end is unaligned)
__global__ void __launch_bounds__(256) k_copySegments(CopySegs segs)
{
    const CopySeg sg = segs.s[blockIdx.y];
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    if (sg.kind == SEG_PACK_XYZ) {
        const float4* s = (const float4*)sg.src;
        float* d = (float*)sg.dst;
        for (size_t i = tid; i < (sg.bytes >> 4); i += nth) {
            const float4 v = s[i];
            d[3 * i + 0] = v.x;
            d[3 * i + 1] = v.y;
            d[3 * i + 2] = v.z;
        }
        return;
    }
    if (sg.kind == SEG_UNPACK_XYZ) {
        const float* s = (const float*)sg.src;
        float4* d = (float4*)sg.dst;
        for (size_t i = tid; i < (sg.bytes >> 4); i += nth) d[i] = make_float4(s[3 * i + 0], s[3 * i + 1], s[3 * i + 2], 0.0f);
        return;
    }
    if ((((unsigned long long)sg.src | (unsigned long long)sg.dst) & 15ull) == 0) {
        const uint4* s = (const uint4*)sg.src;
        uint4* d = (uint4*)sg.dst;
        const size_t n16 = sg.bytes >> 4;
        for (size_t i = tid; i < n16; i += nth) d[i] = s[i];
        const unsigned* s4 = (const unsigned*)sg.src;
        unsigned* d4 = (unsigned*)sg.dst;
        for (size_t i = (n16 << 2) + tid; i < (sg.bytes >> 2); i += nth) d4[i] = s4[i];
    } else {
        const unsigned* s4 = (const unsigned*)sg.src;
        unsigned* d4 = (unsigned*)sg.dst;
        for (size_t i = tid; i < (sg.bytes >> 2); i += nth) d4[i] = s4[i];
    }
}

__global__ void __launch_bounds__(256) k_addInPlace(float* __restrict__ a, const float* __restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = a[i] + b[i];
}

// v[c] = max(v[c], others[p * count + c]) for p != me
__global__ void k_maxInts(int* v, const int* others, int world, int me, int count)
{
    if ((int)threadIdx.x < count && blockIdx.x == 0) {
        const int c = threadIdx.x;
        int m = v[c];
        for (int p = 0; p < world; p++)
            if (p != me && others[p * count + c] > m) m = others[p * count + c];
        v[c] = m;
    }
}

int copy_segments(const std::vector<CopySeg>& segs, hipStream_t st)
{
    for (size_t i = 0; i < segs.size(); i += kSegsPerLaunch) {
        CopySegs a;
        memset((void*)&a, 0, sizeof(a));
        const int n = (int)((segs.size() - i < (size_t)kSegsPerLaunch) ? segs.size() - i : kSegsPerLaunch);
        for (int j = 0; j < n; j++) a.s[j] = segs[i + j];
        hipLaunchKernelGGL(k_copySegments, dim3(kBlocksPerSeg, n), dim3(256), 0, st, a);
        D_HIP(hipGetLastError());
    }
    return MFSR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// transport table

struct Transport {
    int rank = 0, world = 1;
    virtual ~Transport() {}
    virtual const char* name() const = 0;
    // point-to-point calls between groupStart and groupEnd complete together (ncclGroupStart/End semantics): none of them
    // blocks on a peer before groupEnd.  Every call of one group uses the same stream.  A failing call is remembered: the
    // caller keeps the first error and ALWAYS reaches groupEnd.
    virtual int groupStart() = 0;
    virtual int groupEnd() = 0;
    virtual int send(const void* buf, size_t bytes, int peer, hipStream_t st) = 0;
    virtual int recv(void* buf, size_t bytes, int peer, hipStream_t st) = 0;
    // collectives (called outside groups)
    virtual int allReduceMaxI32(int* buf, int count, hipStream_t st) = 0;               // count <= 2 ints, in place
    virtual int reduceSumF32(float* buf, size_t count, int root, hipStream_t st) = 0;   // in place on root
    virtual int reduceScatterSumF32(float* buf, size_t chunk, hipStream_t st) = 0;      // buf = world chunks; mine summed in place
    // tells the peers that this rank gave up (local back end: wakes their host-side waits); no-op for RCCL
    virtual void abort() {}
};

// ---- RCCL: one process per GPU -----------------------------------------------------------------------------------------
struct RcclTransport : Transport {
    ncclComm_t comm = nullptr;
    const char* name() const override { return "rccl"; }
    static int nccl(ncclResult_t r, const char* what)
    {
        if (r == ncclSuccess) return MFSR_OK;
        fprintf(stderr, "mfsr_dist: %s failed: %s\n", what, ncclGetErrorString(r));
        return MFSR_E_COMM;
    }
    int init(int worldSize, int rank_, const void* id)
    {
        rank = rank_;
        world = worldSize;
        ncclUniqueId u;
        memcpy(&u, id, sizeof(u));
        return nccl(ncclCommInitRank(&comm, worldSize, u, rank_), "ncclCommInitRank");
    }
    ~RcclTransport() override
    {
        if (comm) (void)ncclCommDestroy(comm);
    }
    int groupStart() override { return nccl(ncclGroupStart(), "ncclGroupStart"); }
    int groupEnd() override { return nccl(ncclGroupEnd(), "ncclGroupEnd"); }
    int send(const void* buf, size_t bytes, int peer, hipStream_t st) override
    {
        return nccl(ncclSend(buf, bytes, ncclUint8, peer, comm, st), "ncclSend");
    }
    int recv(void* buf, size_t bytes, int peer, hipStream_t st) override
    {
        return nccl(ncclRecv(buf, bytes, ncclUint8, peer, comm, st), "ncclRecv");
    }
    int allReduceMaxI32(int* buf, int count, hipStream_t st) override
    {
        return nccl(ncclAllReduce(buf, buf, (size_t)count, ncclInt32, ncclMax, comm, st), "ncclAllReduce");
    }
    int reduceSumF32(float* buf, size_t count, int root, hipStream_t st) override
    {
        return nccl(ncclReduce(buf, buf, count, ncclFloat, ncclSum, root, comm, st), "ncclReduce");
    }
    int reduceScatterSumF32(float* buf, size_t chunk, hipStream_t st) override
    {
        return nccl(ncclReduceScatter(buf, buf + chunk * (size_t)rank, chunk, ncclFloat, ncclSum, comm, st), "ncclReduceScatter");
    }
};

// ---- local: G ranks in one process, one host thread per rank -----------------------------------------------------------
//
// A message is a PUSH: the receiver posts (dst, bytes, "dst may be written" event), the sender waits on the host for the
// post, makes its stream wait for that event, enqueues the peer copy on its own stream (writes over xGMI), records a
// "sent" event, and the receiver's stream waits for it.  Per group and rank:
//   1. record evReady on my stream; post every receive of the group into mailbox[src][me]
//   2. for every send: wait (host) for the peer's post, stream-wait its ready event, enqueue the copy; record evSent;
//      mark the messages sent
//   3. for every receive: wait (host) until sent, stream-wait the sender's evSent; mark acknowledged
//   4. for every send: wait (host) until acknowledged -- only then may evSent / evReady be recorded again
// Sends and receives between one pair match in program order, as with RCCL.  Host waits time out (MFSR_DIST_TIMEOUT_S,
// default 60 s) and a failing rank marks the world failed, which wakes every waiter: nothing hangs.
struct LocalMsg {
    void* dst = nullptr;
    size_t bytes = 0;
    int dstDev = 0;
    hipEvent_t ready = nullptr, sent = nullptr;
    int state = 0;  // 0 posted, 1 copy enqueued + `sent` recorded, 2 the receiver's stream waits for `sent`
};

struct LocalWorld {
    std::mutex mu;
    std::condition_variable cv;
    int G = 0;
    std::vector<int> dev;
    std::vector<std::deque<std::shared_ptr<LocalMsg>>> box;  // [src * G + dst]: receives posted by dst, waiting for src
    bool failed = false;
    double timeoutSec = 60.0;
    template <class Pred>
    bool wait(std::unique_lock<std::mutex>& lk, Pred pred)
    {
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeoutSec);
        while (!failed && !pred()) {
            if (cv.wait_until(lk, deadline) == std::cv_status::timeout && !pred()) {
                fprintf(stderr, "mfsr_dist(local): a peer did not arrive within %.0f s\n", timeoutSec);
                failed = true;
                cv.notify_all();
                return false;
            }
        }
        return !failed;
    }
};

struct LocalTransport : Transport {
    LocalWorld* w = nullptr;
    int dev = 0;
    hipEvent_t evReady = nullptr, evSent = nullptr;
    bool inGroup = false;
    hipStream_t gStream = nullptr;
    bool gStreamSet = false;
    struct PSend {
        const void* src;
        size_t bytes;
        int peer;
        std::shared_ptr<LocalMsg> m;
    };
    struct PRecv {
        std::shared_ptr<LocalMsg> m;
        int peer;
    };
    std::vector<PSend> sends;
    std::vector<PRecv> recvs;
    float* scratch = nullptr;
    size_t scratchBytes = 0;
    int* iscratch = nullptr;  // world ints (workspace)

    const char* name() const override { return "local"; }
    int init(LocalWorld* world_, int rank_, int dev_, int* iscratch_)
    {
        w = world_;
        rank = rank_;
        world = world_->G;
        dev = dev_;
        iscratch = iscratch_;
        D_HIP(hipEventCreateWithFlags(&evReady, hipEventDisableTiming));
        D_HIP(hipEventCreateWithFlags(&evSent, hipEventDisableTiming));
        return MFSR_OK;
    }
    ~LocalTransport() override
    {
        if (evReady) (void)hipEventDestroy(evReady);
        if (evSent) (void)hipEventDestroy(evSent);
        if (scratch) (void)hipFree(scratch);
    }
    void abort() override
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->failed = true;
        w->cv.notify_all();
    }
    int fail(const char* what)
    {
        fprintf(stderr, "mfsr_dist(local) rank %d: %s\n", rank, what);
        abort();
        return MFSR_E_COMM;
    }
    int useStream(hipStream_t st)
    {
        if (gStreamSet && gStream != st) return fail("every call of one group must use the same stream");
        gStream = st;
        gStreamSet = true;
        return MFSR_OK;
    }
    int groupStart() override
    {
        if (inGroup) return fail("nested groupStart");
        inGroup = true;
        gStreamSet = false;
        sends.clear();
        recvs.clear();
        return MFSR_OK;
    }
    int send(const void* buf, size_t bytes, int peer, hipStream_t st) override
    {
        const bool single = !inGroup;
        if (single) D_TRY(groupStart());
        int rc = (peer < 0 || peer >= world || peer == rank || !buf) ? fail("send: bad peer / buffer") : useStream(st);
        if (rc == MFSR_OK) sends.push_back(PSend{buf, bytes, peer, nullptr});
        if (single) D_KEEP(rc, groupEnd());
        return rc;
    }
    int recv(void* buf, size_t bytes, int peer, hipStream_t st) override
    {
        const bool single = !inGroup;
        if (single) D_TRY(groupStart());
        int rc = (peer < 0 || peer >= world || peer == rank || !buf) ? fail("recv: bad peer / buffer") : useStream(st);
        if (rc == MFSR_OK) {
            auto m = std::make_shared<LocalMsg>();
            m->dst = buf;
            m->bytes = bytes;
            m->dstDev = dev;
            m->ready = evReady;
            recvs.push_back(PRecv{m, peer});
        }
        if (single) D_KEEP(rc, groupEnd());
        return rc;
    }
    int groupEnd() override
    {
        if (!inGroup) return fail("groupEnd without groupStart");
        inGroup = false;
        if (sends.empty() && recvs.empty()) return MFSR_OK;
        hipStream_t st = gStream;
        int rc = MFSR_OK;
        const int G = world;
        // 1. my receive buffers are free once my stream gets here
        if (!recvs.empty()) {
            hipError_t e = hipEventRecord(evReady, st);
            if (e != hipSuccess) return fail(hipGetErrorString(e));
        }
        {
            std::unique_lock<std::mutex> lk(w->mu);
            if (w->failed) return MFSR_E_COMM;
            for (auto& r : recvs) w->box[(size_t)r.peer * G + rank].push_back(r.m);
            w->cv.notify_all();
            // 2. my sends, in program order
            for (auto& s : sends) {
                auto& q = w->box[(size_t)rank * G + s.peer];
                if (!w->wait(lk, [&] { return !q.empty(); })) return MFSR_E_COMM;
                s.m = q.front();
                q.pop_front();
                if (s.m->bytes != s.bytes) {
                    fprintf(stderr, "mfsr_dist(local): rank %d sends %zu bytes to rank %d, which expects %zu\n", rank, s.bytes, s.peer,
                            s.m->bytes);
                    w->failed = true;
                    w->cv.notify_all();
                    return MFSR_E_COMM;
                }
            }
        }
        for (auto& s : sends) {
            hipError_t e = hipStreamWaitEvent(st, s.m->ready, 0);
            if (e == hipSuccess && s.bytes) {
                e = (s.m->dstDev == dev) ? hipMemcpyAsync(s.m->dst, s.src, s.bytes, hipMemcpyDeviceToDevice, st)
                                         : hipMemcpyPeerAsync(s.m->dst, s.m->dstDev, s.src, dev, s.bytes, st);
            }
            if (e != hipSuccess) return fail(hipGetErrorString(e));
        }
        if (!sends.empty()) {
            hipError_t e = hipEventRecord(evSent, st);
            if (e != hipSuccess) return fail(hipGetErrorString(e));
        }
        {
            std::unique_lock<std::mutex> lk(w->mu);
            for (auto& s : sends) {
                s.m->sent = evSent;
                s.m->state = 1;
            }
            w->cv.notify_all();
            // 3. my receives have been written once their senders' streams pass `sent`
            for (auto& r : recvs) {
                if (!w->wait(lk, [&] { return r.m->state >= 1; })) return MFSR_E_COMM;
                lk.unlock();
                hipError_t e = hipStreamWaitEvent(st, r.m->sent, 0);
                lk.lock();
                if (e != hipSuccess) {
                    w->failed = true;
                    w->cv.notify_all();
                    fprintf(stderr, "mfsr_dist(local): hipStreamWaitEvent failed: %s\n", hipGetErrorString(e));
                    return MFSR_E_COMM;
                }
                r.m->state = 2;
            }
            w->cv.notify_all();
            // 4. my events may be recorded again once every receiver's stream holds its wait
            for (auto& s : sends)
                if (!w->wait(lk, [&] { return s.m->state >= 2; })) return MFSR_E_COMM;
        }
        sends.clear();
        recvs.clear();
        return rc;
    }
    int ensureScratch(size_t bytes)
    {
        if (bytes <= scratchBytes) return MFSR_OK;
        if (scratch) D_HIP(hipFree(scratch));
        scratch = nullptr;
        scratchBytes = 0;
        D_HIP(hipMalloc((void**)&scratch, bytes));
        scratchBytes = bytes;
        return MFSR_OK;
    }
    static void add(float* a, const float* b, size_t n, hipStream_t st)
    {
        size_t blocks = (n + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_addInPlace, dim3((unsigned)blocks), dim3(256), 0, st, a, b, n);
    }
    int allReduceMaxI32(int* buf, int count, hipStream_t st) override
    {
        if (world == 1) return MFSR_OK;
        if (count < 1 || count > 2) return fail("allReduceMaxI32: count must be 1 or 2");
        int rc = groupStart();
        if (rc != MFSR_OK) return rc;
        for (int p = 0; p < world; p++) {
            if (p == rank) continue;
            D_KEEP(rc, send(buf, sizeof(int) * count, p, st));
            D_KEEP(rc, recv(iscratch + p * count, sizeof(int) * count, p, st));
        }
        D_KEEP(rc, groupEnd());
        if (rc != MFSR_OK) return rc;
        hipLaunchKernelGGL(k_maxInts, dim3(1), dim3(64), 0, st, buf, (const int*)iscratch, world, rank, count);
        D_HIP(hipGetLastError());
        return MFSR_OK;
    }
    // root adds the peers' buffers in rank order (deterministic; RCCL's order is its own)
    int reduceSumF32(float* buf, size_t count, int root, hipStream_t st) override
    {
        if (world == 1) return MFSR_OK;
        if (rank != root) return send(buf, count * sizeof(float), root, st);
        int rc = ensureScratch(count * sizeof(float));
        if (rc != MFSR_OK) {
            abort();
            return rc;
        }
        for (int p = 0; p < world; p++) {
            if (p == root) continue;
            D_TRY(recv(scratch, count * sizeof(float), p, st));
            add(buf, scratch, count, st);
            D_HIP(hipGetLastError());
        }
        return MFSR_OK;
    }
    int reduceScatterSumF32(float* buf, size_t chunk, hipStream_t st) override
    {
        if (world == 1) return MFSR_OK;
        int rc = ensureScratch(chunk * sizeof(float) * (size_t)(world - 1));
        if (rc != MFSR_OK) {
            abort();
            return rc;
        }
        rc = groupStart();
        if (rc != MFSR_OK) return rc;
        int slot = 0;
        for (int p = 0; p < world; p++) {
            if (p == rank) continue;
            D_KEEP(rc, send(buf + chunk * (size_t)p, chunk * sizeof(float), p, st));
            D_KEEP(rc, recv(scratch + chunk * (size_t)slot++, chunk * sizeof(float), p, st));
        }
        D_KEEP(rc, groupEnd());
        if (rc != MFSR_OK) return rc;
        for (int s = 0; s < world - 1; s++) add(buf + chunk * (size_t)rank, scratch + chunk * (size_t)s, chunk, st);
        D_HIP(hipGetLastError());
        return MFSR_OK;
    }
};

// ---------------------------------------------------------------------------------------------------------------------

struct DistLayout {
    int W, H, s, N, hrW, hrH, tw, th, hw, hh;
    int flowPitch, maskPitch;
    size_t burstWs, accBytes, rawBytes, flowBytes, maskBytes, out16Bytes;
    size_t sendBytes, recvBytes;  // packed exchange buffers, sized for whole raw frames (the largest halo)
    size_t offBurst, offBurst2, offImg, offTw, offRaw, offFlow, offMask, offOut16, offFlag, offSend, offRecv, total;
    int sets;  // per-frame product buffer sets and burst contexts: 2 when bursts are pipelined (world > 1), else 1
};

// bytes of one frame's rows inside a packed message for the stripe `pl` (every piece 16-byte aligned)
inline size_t msg_frame_bytes(const DistLayout& L, const mfsr_stripe_plan& pl)
{
    // (certainty rows travel as 3 of their 4 floats)
    return up((size_t)pl.rawRows * L.W * 2, 16) + up((size_t)pl.flowRows * L.flowPitch, 16) + up((size_t)pl.maskRows * L.maskPitch / 4 * 3, 16);
}
inline int frames_of(int N, int rank, int world) { return rank < N ? (N - rank + world - 1) / world : 0; }

int make_layout(const mfsr_config* c, int world, DistLayout* L)
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
    L->flowPitch = L->tw * 8;    // dense rows: a row range is one contiguous piece
    L->maskPitch = L->hw * 16;
    L->burstWs = mfsr_burst_workspace_bytes(c);
    if (L->burstWs == 0) return MFSR_E_INVALID;
    L->accBytes = mfsr_burst_accumulator_bytes(c);
    L->rawBytes = (size_t)L->W * L->H * 2;
    L->flowBytes = (size_t)L->flowPitch * L->th;
    L->maskBytes = (size_t)L->maskPitch * L->hh;
    L->out16Bytes = (size_t)L->hrW * L->hrH * 6;
    // packed exchange: worst case over the ranks, with whole raw frames (rawHalo >= H)
    if (world > 1) {
        size_t sendMax = 0, recvMax = 0;
        std::vector<mfsr_stripe_plan> pl(world);
        for (int p = 0; p < world; p++)
            if (mfsr_dist_stripe_plan(c, world, p, L->H + 4, &pl[p]) != MFSR_OK) return MFSR_E_INVALID;
        for (int r = 0; r < world; r++) {
            size_t snd = 0, rcv = 0;
            for (int p = 0; p < world; p++) {
                if (p == r) continue;
                if (pl[p].rowEnd > pl[p].rowBegin) snd += up((size_t)frames_of(L->N, r, world) * msg_frame_bytes(*L, pl[p]), 256);
                if (pl[r].rowEnd > pl[r].rowBegin) rcv += up((size_t)frames_of(L->N, p, world) * msg_frame_bytes(*L, pl[r]), 256);
            }
            if (snd > sendMax) sendMax = snd;
            if (rcv > recvMax) recvMax = rcv;
        }
        L->sendBytes = sendMax;
        L->recvBytes = recvMax;
    }
    size_t off = 0;
    auto take = [&](size_t bytes) {
        off = up(off, 256);
        const size_t o = off;
        off += bytes;
        return o;
    };
    L->sets = world > 1 ? 2 : 1;
    L->offBurst = take(L->burstWs);
    L->offBurst2 = L->sets > 1 ? take(L->burstWs) : L->offBurst;
    L->offImg = take(L->accBytes);
    L->offTw = take(L->accBytes);
    L->offRaw = take(up(L->rawBytes, 256) * L->N * L->sets);
    L->offFlow = take(up(L->flowBytes, 256) * L->N * L->sets);
    L->offMask = take(up(L->maskBytes, 256) * L->N * L->sets);
    L->offOut16 = take(L->out16Bytes);
    L->offFlag = take(256 + 2 * sizeof(int) * (size_t)(world > 64 ? world : 64));
    L->offSend = take(L->sendBytes);
    L->offRecv = take(L->recvBytes);
    L->total = up(off, 256);
    return MFSR_OK;
}

}  // namespace

struct mfsr_dist {
    mfsr_config cfg;
    DistLayout L;
    int rank, world, rawHalo, device;
    Transport* T;
    mfsr_burst* burst;
    mfsr_burst* burst2;  // the second context of pipelined bursts (burst i works in context i & 1); == burst when not pipelined
    char* base;
    mfsr_float3 *imgOut, *totalWeights;
    uint16_t* out16;
    int* flag;   // three (status, max-flow bits) pairs used in turn (see process_stripes); the local transport's scratch follows at +8
    // STRIPES: every transport call goes to commStream, in the same order on every rank (exchange i, gather i, exchange i+1, ..),
    // event-linked to the caller's stream, so that the gather of burst i overlaps the alignment of burst i+1
    hipStream_t commStream;
    hipEvent_t evAligned, evExchanged, evExchanged2, evFinished, evGathered, evBackStart;
    int* lastFlag;  // (status, bits of the measured max |vertical flow|) of the last burst whose back half was enqueued
    // STRIPES, pipelined (world > 1, MFSR_DIST_PIPELINE != 0): process_burst(i) runs the reference products, the alignment and
    // the exchange of burst i, then the fuse / finish / gather of burst i - 1 -- whose exchange completed long ago -- so the
    // caller's stream never waits for an exchange; the last burst's back half runs in mfsr_dist_wait_output
    int pipeline;
    struct PendingBack {
        bool has;
        int ctx, rawHalo;
        std::vector<const uint16_t*>* raws;
        uint16_t* out16;
        int* status;
        int* flag;
    } back;
    bool gatherPending;
    int overlap;
    long long burstNo;
    long long messagesSent, bytesSent;  // exchange + gather of the last burst (mfsr_dist_exchange_stats)
    int cur;  // product buffer set / burst context of the burst being enqueued
    uint16_t* raw(int k) const { return (uint16_t*)(base + L.offRaw + up(L.rawBytes, 256) * ((size_t)cur * L.N + k)); }
    mfsr_float2* flow(int k) const { return (mfsr_float2*)(base + L.offFlow + up(L.flowBytes, 256) * ((size_t)cur * L.N + k)); }
    mfsr_float4* mask(int k) const { return (mfsr_float4*)(base + L.offMask + up(L.maskBytes, 256) * ((size_t)cur * L.N + k)); }
    mfsr_burst* ctx() const { return cur ? burst2 : burst; }
    char* sendBuf() const { return base + L.offSend; }
    char* recvBuf() const { return base + L.offRecv; }
};

extern "C" int mfsr_dist_get_unique_id(void* id)
{
    D_REQUIRE(id != nullptr);
    ncclUniqueId u;
    ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) {
        fprintf(stderr, "mfsr_dist: ncclGetUniqueId failed: %s\n", ncclGetErrorString(r));
        return MFSR_E_COMM;
    }
    memcpy(id, &u, sizeof(u));
    return MFSR_OK;
}

extern "C" size_t mfsr_dist_workspace_bytes(const mfsr_config* cfg, int worldSize)
{
    DistLayout L;
    if (!cfg || worldSize < 1 || make_layout(cfg, worldSize, &L) != MFSR_OK) return 0;
    return L.total;
}

static void dist_free(mfsr_dist* d)
{
    if (!d) return;
    if (d->commStream) (void)hipStreamSynchronize(d->commStream);
    delete d->T;
    if (d->evAligned) (void)hipEventDestroy(d->evAligned);
    if (d->evExchanged) (void)hipEventDestroy(d->evExchanged);
    if (d->evExchanged2) (void)hipEventDestroy(d->evExchanged2);
    if (d->evFinished) (void)hipEventDestroy(d->evFinished);
    if (d->evGathered) (void)hipEventDestroy(d->evGathered);
    if (d->evBackStart) (void)hipEventDestroy(d->evBackStart);
    if (d->commStream) (void)hipStreamDestroy(d->commStream);
    if (d->burst2 && d->burst2 != d->burst) mfsr_burst_destroy(d->burst2);
    if (d->burst) mfsr_burst_destroy(d->burst);
    delete d->back.raws;
    delete d;
}

// everything of a context except its transport, on the CURRENT device
static int dist_new(mfsr_dist** out, const mfsr_config* cfg, int rank, int worldSize, void* workspace, size_t workspaceBytes)
{
    D_REQUIRE(out && cfg && workspace && worldSize >= 1 && rank >= 0 && rank < worldSize);
    D_REQUIRE(((uintptr_t)workspace & 255) == 0);
    mfsr_dist* d = new (std::nothrow) mfsr_dist;
    D_REQUIRE(d != nullptr);
    memset((void*)d, 0, sizeof(*d));
    d->cfg = *cfg;
    int rc = make_layout(cfg, worldSize, &d->L);
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
    hipError_t he = hipGetDevice(&d->device);
    if (he == hipSuccess) rc = mfsr_burst_create(&d->burst, cfg, d->base + d->L.offBurst, d->L.burstWs);
    d->burst2 = d->burst;
    if (he == hipSuccess && rc == MFSR_OK && d->L.sets > 1)
        rc = mfsr_burst_create(&d->burst2, cfg, d->base + d->L.offBurst2, d->L.burstWs);
    d->back.raws = new (std::nothrow) std::vector<const uint16_t*>();
    if (!d->back.raws && rc == MFSR_OK) rc = MFSR_E_INVALID;
    if (he != hipSuccess || rc != MFSR_OK) {
        dist_free(d);
        return he != hipSuccess ? (int)he : rc;
    }
    const char* e = getenv("MFSR_DIST_OVERLAP");
    d->overlap = (e && e[0] == '0') ? 0 : 1;
    const char* ep = getenv("MFSR_DIST_PIPELINE");
    d->pipeline = (d->L.sets > 1 && d->overlap && !(ep && ep[0] == '0')) ? 1 : 0;
    {
        // The comm stream takes the HIGHEST priority: RCCL's send / receive / reduce are kernels, and a warp+fuse launch fills
        // every SIMD's register file (four 128-VGPR workgroups per CU) -- a transport kernel on a default-priority queue would
        // wait for the whole launch (0.1 ms per stripe launch, 0.9 ms per whole-frame launch: the tracker launch of round 3's
        // timeline did).  With priority its few workgroups take the first slots the retiring fuse workgroups free (one fuse
        // workgroup lives ~30 us).  The local back end's messages are peer copies (copy engines): unaffected.
        int lo = 0, hi = 0;
        he = hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (he == hipSuccess) he = hipStreamCreateWithPriority(&d->commStream, hipStreamNonBlocking, hi);
    }
    if (he == hipSuccess) he = hipEventCreateWithFlags(&d->evBackStart, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&d->evAligned, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&d->evExchanged, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&d->evExchanged2, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&d->evFinished, hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&d->evGathered, hipEventDisableTiming);
    if (he != hipSuccess) {
        fprintf(stderr, "mfsr_dist: stream / event creation failed: %s\n", hipGetErrorString(he));
        dist_free(d);  // destroys whatever was created (the struct was zeroed)
        return (int)he;
    }
    *out = d;
    return MFSR_OK;
}

extern "C" int mfsr_dist_create(mfsr_dist** out, const mfsr_config* cfg, int rank, int worldSize, const void* id, void* workspace,
                                size_t workspaceBytes)
{
    D_REQUIRE(out && id);
    mfsr_dist* d = nullptr;
    D_TRY(dist_new(&d, cfg, rank, worldSize, workspace, workspaceBytes));
    RcclTransport* t = new (std::nothrow) RcclTransport;
    int rc = t ? t->init(worldSize, rank, id) : MFSR_E_INVALID;
    if (rc != MFSR_OK) {
        delete t;
        dist_free(d);
        return rc;
    }
    d->T = t;
    *out = d;
    return MFSR_OK;
}

extern "C" void mfsr_dist_destroy(mfsr_dist* d) { dist_free(d); }

extern "C" mfsr_burst* mfsr_dist_burst(mfsr_dist* d) { return d ? d->burst : nullptr; }

extern "C" const char* mfsr_dist_transport(const mfsr_dist* d) { return (d && d->T) ? d->T->name() : ""; }

extern "C" int mfsr_dist_set_raw_halo(mfsr_dist* d, int rawHalo)
{
    D_REQUIRE(d && rawHalo >= 4 && rawHalo <= 65536);
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

extern "C" int mfsr_dist_exchange_stats(const mfsr_dist* d, long long* messagesSent, long long* bytesSent)
{
    D_REQUIRE(d != nullptr);
    if (messagesSent) *messagesSent = d->messagesSent;
    if (bytesSent) *bytesSent = d->bytesSent;
    return MFSR_OK;
}

// rank 0 collects the u16 stripes (stripe p = HR rows [b_p, e_p) of rank p's staging image): world - 1 messages into rank 0
static int gather_stripes(mfsr_dist* d, const std::vector<mfsr_stripe_plan>& plan, uint16_t* out16, hipStream_t st)
{
    const size_t rowBytes = (size_t)d->L.hrW * 6;
    int rc = d->T->groupStart();
    if (rc != MFSR_OK) return rc;
    if (d->rank == 0) {
        for (int p = 1; p < d->world && rc == MFSR_OK; p++) {
            const size_t n = (size_t)(plan[p].rowEnd - plan[p].rowBegin) * rowBytes;
            if (n) D_KEEP(rc, d->T->recv((char*)out16 + (size_t)plan[p].rowBegin * rowBytes, n, p, st));
        }
    } else {
        const size_t n = (size_t)(plan[d->rank].rowEnd - plan[d->rank].rowBegin) * rowBytes;
        if (n) {
            D_KEEP(rc, d->T->send((const char*)d->out16 + (size_t)plan[d->rank].rowBegin * rowBytes, n, 0, st));
            d->messagesSent++;
            d->bytesSent += (long long)n;
        }
    }
    D_KEEP(rc, d->T->groupEnd());
    return rc;
}

// The STRIPES exchange: to peer p the rows of MY frames' raw / flow / certainty that p's stripe reads, from the owner of
// frame k the rows MY stripe reads.  One packed message per peer and direction: a pack launch gathers the row ranges
// (three per frame) into the send buffer, one send + one receive per peer inside one group, an unpack launch scatters the
// received rows into the full-size per-frame buffers the fuse reads (frames in ascending k inside a message).
static int exchange_rows(mfsr_dist* d, const std::vector<mfsr_stripe_plan>& plan, const std::vector<const uint16_t*>& mineRaw,
                         hipStream_t st)
{
    const DistLayout& L = d->L;
    const int N = d->cfg.frames, G = d->world, me = d->rank;
    const mfsr_stripe_plan& mine = plan[me];
    const bool iFuse = mine.rowEnd > mine.rowBegin;
    std::vector<CopySeg> pack, unpack;
    std::vector<size_t> sendOff(G, 0), sendLen(G, 0), recvOff(G, 0), recvLen(G, 0);
    size_t so = 0, ro = 0;
    for (int p = 0; p < G; p++) {
        if (p == me) continue;
        if (plan[p].rowEnd > plan[p].rowBegin) {  // p fuses: it needs my frames
            const mfsr_stripe_plan& q = plan[p];
            const size_t rawB = (size_t)q.rawRows * L.W * 2, flowB = (size_t)q.flowRows * L.flowPitch, maskB = (size_t)q.maskRows * L.maskPitch;
            const size_t maskMsg = maskB / 4 * 3;  // .xyz of every float4 texel
            sendOff[p] = so;
            char* dst = d->sendBuf() + so;
            for (int k = me; k < N; k += G) {
                pack.push_back(CopySeg{(const char*)mineRaw[k] + (size_t)q.rawRow0 * L.W * 2, dst, rawB});
                dst += up(rawB, 16);
                pack.push_back(CopySeg{(const char*)d->flow(k) + (size_t)q.flowRow0 * L.flowPitch, dst, flowB});
                dst += up(flowB, 16);
                pack.push_back(CopySeg{(const char*)d->mask(k) + (size_t)q.maskRow0 * L.maskPitch, dst, maskB, SEG_PACK_XYZ});
                dst += up(maskMsg, 16);
            }
            sendLen[p] = (size_t)(dst - (d->sendBuf() + so));
            so += up(sendLen[p], 256);
        }
        if (iFuse) {  // I fuse: I need p's frames
            const size_t rawB = (size_t)mine.rawRows * L.W * 2, flowB = (size_t)mine.flowRows * L.flowPitch,
                         maskB = (size_t)mine.maskRows * L.maskPitch, maskMsg = maskB / 4 * 3;
            recvOff[p] = ro;
            const char* src = d->recvBuf() + ro;
            for (int k = p; k < N; k += G) {
                unpack.push_back(CopySeg{src, (char*)d->raw(k) + (size_t)mine.rawRow0 * L.W * 2, rawB});
                src += up(rawB, 16);
                unpack.push_back(CopySeg{src, (char*)d->flow(k) + (size_t)mine.flowRow0 * L.flowPitch, flowB});
                src += up(flowB, 16);
                unpack.push_back(CopySeg{src, (char*)d->mask(k) + (size_t)mine.maskRow0 * L.maskPitch, maskB, SEG_UNPACK_XYZ});
                src += up(maskMsg, 16);
            }
            recvLen[p] = (size_t)(src - (d->recvBuf() + ro));
            ro += up(recvLen[p], 256);
        }
    }
    if (so > L.sendBytes || ro > L.recvBytes) {
        fprintf(stderr, "mfsr_dist: packed exchange buffers too small (%zu / %zu of %zu / %zu bytes)\n", so, ro, L.sendBytes, L.recvBytes);
        return MFSR_E_WORKSPACE;
    }
    D_TRY(copy_segments(pack, st));
    int rc = d->T->groupStart();
    if (rc != MFSR_OK) return rc;
    for (int p = 0; p < G && rc == MFSR_OK; p++) {
        if (p == me) continue;
        if (sendLen[p]) {
            D_KEEP(rc, d->T->send(d->sendBuf() + sendOff[p], sendLen[p], p, st));
            d->messagesSent++;
            d->bytesSent += (long long)sendLen[p];
        }
        if (recvLen[p]) D_KEEP(rc, d->T->recv(d->recvBuf() + recvOff[p], recvLen[p], p, st));
    }
    D_KEEP(rc, d->T->groupEnd());
    if (rc != MFSR_OK) return rc;
    return copy_segments(unpack, st);
}

// back half of a STRIPES burst whose products sit in set `ctx`: flow-bound check, every frame fused in frame order onto this
// rank's HR rows, finish, gather of the u16 stripes onto rank 0, status
static int stripes_back(mfsr_dist* d, int ctx, int rawHalo, const std::vector<const uint16_t*>& raws, uint16_t* out16, int* status, int* flag,
                        hipStream_t st)
{
    const mfsr_config& c = d->cfg;
    const DistLayout& L = d->L;
    const int N = c.frames, G = d->world, me = d->rank;
    std::vector<mfsr_stripe_plan> plan(G);
    for (int p = 0; p < G; p++) D_TRY(mfsr_dist_stripe_plan(&c, G, p, rawHalo, &plan[p]));
    const mfsr_stripe_plan& mine = plan[me];
    hipStream_t B = d->overlap ? d->commStream : st;
    const int saved = d->cur;
    d->cur = ctx;
    struct Restore {
        mfsr_dist* d;
        int v;
        ~Restore() { d->cur = v; }
    } restore{d, saved};
    D_HIP(hipStreamWaitEvent(st, ctx ? d->evExchanged2 : d->evExchanged, 0));
    // Rank 0 posts the receives of the peers' u16 stripes BEFORE its own fuse: their rows of out16 are disjoint from rank 0's
    // stripe, so a peer that finishes first sends at once and the 7/8 of the image that come over the links (the largest term
    // of the comm chain at configs[2]) travel while rank 0 is still fusing.  Ordered after everything the caller enqueued
    // before this burst's back half (out16 may still be read by it) and after the previous gather (same stream B).
    const bool earlyGather = G > 1 && me == 0 && d->overlap;
    if (earlyGather) {
        D_HIP(hipEventRecord(d->evBackStart, st));
        D_HIP(hipStreamWaitEvent(B, d->evBackStart, 0));
        D_TRY(gather_stripes(d, plan, out16, B));
    }
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
            D_TRY(mfsr_burst_fuse_rows(d->ctx(), n, rg, fg, L.flowPitch, mg, L.maskPitch, d->imgOut, d->totalWeights, k == 0 ? 1 : 0,
                                       mine.rowBegin, mine.rowEnd, (mfsr_stream_t)st));
        }
        uint16_t* dst = me == 0 ? out16 : d->out16;
        // the staging image may still be on its way to rank 0 (gather of the previous burst, on B)
        if (d->gatherPending) D_HIP(hipStreamWaitEvent(st, d->evGathered, 0));
        D_TRY(mfsr_burst_finish_rows(d->ctx(), d->imgOut, d->totalWeights, nullptr, dst, mine.rowBegin, mine.rowEnd - mine.rowBegin,
                                     (mfsr_stream_t)st));
    }
    // gather + status on B after the finish on A; the caller's stream does NOT wait for it (mfsr_dist_wait_output does):
    // the next burst's alignment runs meanwhile
    D_HIP(hipEventRecord(d->evFinished, st));
    D_HIP(hipStreamWaitEvent(B, d->evFinished, 0));
    if (G > 1) {
        if (!earlyGather) D_TRY(gather_stripes(d, plan, out16, B));
        D_TRY(d->T->allReduceMaxI32(flag, 2, B));  // (status, measured max |vertical flow|)
    }
    if (status) D_HIP(hipMemcpyAsync(status, flag, sizeof(int), hipMemcpyDeviceToDevice, B));
    D_HIP(hipEventRecord(d->evGathered, B));
    d->gatherPending = true;
    d->lastFlag = flag;
    return MFSR_OK;
}

// the back half of the burst that is still waiting for it (pipelined STRIPES bursts)
static int drain_back(mfsr_dist* d, hipStream_t st)
{
    if (!d->back.has) return MFSR_OK;
    d->back.has = false;
    return stripes_back(d, d->back.ctx, d->back.rawHalo, *d->back.raws, d->back.out16, d->back.status, d->back.flag, st);
}

static int process_stripes(mfsr_dist* d, const uint16_t* const* frames, uint16_t* out16, int* status, hipStream_t st)
{
    const mfsr_config& c = d->cfg;
    const DistLayout& L = d->L;
    const int N = c.frames, G = d->world, me = d->rank, ref = c.reference;
    std::vector<mfsr_stripe_plan> plan(G);
    for (int p = 0; p < G; p++) D_TRY(mfsr_dist_stripe_plan(&c, G, p, d->rawHalo, &plan[p]));
    // (status, bits of max |vertical flow|); three pairs in turn: burst i - 3's all-reduce on the comm stream completed before
    // the caller's stream passed the finish of burst i - 2 (it waited for that gather event there), i.e. before this memset
    int* flag = d->flag + 2 * (d->burstNo % 3);
    d->cur = d->pipeline ? (int)(d->burstNo & 1) : 0;
    d->burstNo++;
    d->messagesSent = d->bytesSent = 0;
    // A = the caller's stream (kernels), B = the comm stream (every transport call); with overlap off B = A
    hipStream_t B = d->overlap ? d->commStream : st;
    D_HIP(hipMemsetAsync(flag, 0, 2 * sizeof(int), st));

    // front half: reference products on every rank -- what the alignment reads in full, the kernel parameters and the
    // fallback image for this rank's stripe only (nobody else reads them here) -- then this rank's frames: alignment only
    {
        const mfsr_stripe_plan& mine = plan[me];
        const bool fuses = mine.rowEnd > mine.rowBegin;
        D_TRY(mfsr_burst_set_reference_rows(d->ctx(), frames[ref], fuses ? mine.rowBegin : 0, fuses ? mine.rowEnd : (L.hrH < 16 ? L.hrH : 16),
                                            (mfsr_stream_t)st));
    }
    std::vector<const uint16_t*> raws(N, nullptr);
    {
        // this rank's frames, aligned in batches (one launch per stage for up to four frames)
        std::vector<const uint16_t*> mineRaw;
        std::vector<int> mineRef;
        std::vector<mfsr_float2*> mineFlow;
        std::vector<mfsr_float4*> mineMask;
        for (int k = 0; k < N; k++) {
            if (k % G != me) {
                raws[k] = d->raw(k);
                continue;
            }
            // pipelined bursts: the fuse of this burst runs during the NEXT call -- it (and the pack on the comm stream) read
            // the library's own copy of the rank's frames, so the caller's buffers are released when `stream` has passed
            // this call, like those of any asynchronous call (N / G frames of 2 B per pixel: ~10 us per 4K frame)
            raws[k] = d->pipeline ? d->raw(k) : frames[k];
            if (d->pipeline) D_HIP(hipMemcpyAsync(d->raw(k), frames[k], L.rawBytes, hipMemcpyDeviceToDevice, st));
            mineRaw.push_back(frames[k]);
            mineRef.push_back(k == ref);
            mineFlow.push_back(d->flow(k));
            mineMask.push_back(d->mask(k));
        }
        if (!mineRaw.empty())
            D_TRY(mfsr_burst_align_frames(d->ctx(), (int)mineRaw.size(), mineRaw.data(), mineRef.data(), mineFlow.data(), L.flowPitch,
                                          mineMask.data(), L.maskPitch, (mfsr_stream_t)st));
        // the largest vertical flow of this rank's frames (every frame has one owner: the all-reduce at the end of the burst
        // makes it the burst's): what mfsr_dist_measured_flow reports, for a caller that sizes the raw-row halo from it
        for (size_t i = 0; i < mineFlow.size(); i++)
            D_TRY(mfsr_maxAbsFlowY(mineFlow[i], L.flowPitch, L.tw, L.th, flag + 1, (mfsr_stream_t)st));
    }
    // exchange (on B, after the alignment on A).  (Receive buffers of this set: their last reader was the fuse of the burst two
    // calls ago, or of the previous one when bursts are not pipelined -- both are on A before the event B waits for here.)
    D_HIP(hipEventRecord(d->evAligned, st));
    D_HIP(hipStreamWaitEvent(B, d->evAligned, 0));
    if (G > 1) D_TRY(exchange_rows(d, plan, raws, B));
    D_HIP(hipEventRecord(d->cur ? d->evExchanged2 : d->evExchanged, B));

    if (!d->pipeline) return stripes_back(d, d->cur, d->rawHalo, raws, out16, status, flag, st);
    // pipelined: now the back half of the PREVIOUS burst (its exchange finished while this burst was being aligned); this
    // burst's waits for the next call or for mfsr_dist_wait_output.  The caller keeps its frames and out16 until then.
    const int rc = drain_back(d, st);
    d->back.has = true;
    d->back.ctx = d->cur;
    d->back.rawHalo = d->rawHalo;
    *d->back.raws = raws;
    d->back.out16 = out16;
    d->back.status = status;
    d->back.flag = flag;
    return rc;
}

static int process_reduce(mfsr_dist* d, const uint16_t* const* frames, int mode, uint16_t* out16, int* status, hipStream_t st)
{
    const mfsr_config& c = d->cfg;
    const DistLayout& L = d->L;
    const int N = c.frames, G = d->world, me = d->rank, ref = c.reference;
    d->messagesSent = d->bytesSent = 0;
    D_TRY(mfsr_burst_begin(d->burst, d->imgOut, d->totalWeights, (mfsr_stream_t)st));
    D_TRY(mfsr_burst_set_reference(d->burst, frames[ref], (mfsr_stream_t)st));
    for (int k = me; k < N; k += G)
        D_TRY(mfsr_burst_add_frame(d->burst, frames[k], k == ref, d->imgOut, d->totalWeights, (mfsr_stream_t)st));
    D_TRY(mfsr_burst_flush(d->burst, (mfsr_stream_t)st));  // the pending frame of an odd shard; zeroes if the shard is empty
    const size_t count = (size_t)L.hrW * L.hrH * 3;
    if (status) D_HIP(hipMemsetAsync(status, 0, sizeof(int), st));
    if (G == 1) return mfsr_burst_finish(d->burst, d->imgOut, d->totalWeights, nullptr, out16, (mfsr_stream_t)st);
    if (mode == MFSR_DIST_REDUCE_SCATTER && (L.hrH % G) == 0) {
        const size_t chunk = count / G;
        const int rows = L.hrH / G;
        D_TRY(d->T->reduceScatterSumF32((float*)d->imgOut, chunk, st));
        D_TRY(d->T->reduceScatterSumF32((float*)d->totalWeights, chunk, st));
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
    D_TRY(d->T->reduceSumF32((float*)d->imgOut, count, 0, st));
    D_TRY(d->T->reduceSumF32((float*)d->totalWeights, count, 0, st));
    if (me == 0) return mfsr_burst_finish(d->burst, d->imgOut, d->totalWeights, nullptr, out16, (mfsr_stream_t)st);
    return MFSR_OK;
}

static int check_burst_args(const mfsr_dist* d, const uint16_t* const* frames, int mode, const uint16_t* out16)
{
    D_REQUIRE(frames != nullptr);
    D_REQUIRE(mode == MFSR_DIST_STRIPES || mode == MFSR_DIST_REDUCE || mode == MFSR_DIST_REDUCE_SCATTER);
    D_REQUIRE(frames[d->cfg.reference] != nullptr);
    D_REQUIRE(d->rank != 0 || out16 != nullptr);
    for (int k = d->rank; k < d->cfg.frames; k += d->world) D_REQUIRE(frames[k] != nullptr);
    return MFSR_OK;
}

extern "C" int mfsr_dist_process_burst(mfsr_dist* d, const uint16_t* const* frames, int mode, uint16_t* out16, int* status,
                                       mfsr_stream_t stream)
{
    D_REQUIRE(d && d->T);
    // every argument is checked BEFORE the first enqueue: a rank that returned from the middle of a burst would leave its
    // peers waiting in their transport calls.  (Local back end: a failing rank also wakes its peers' host-side waits.)
    int rc = check_burst_args(d, frames, mode, out16);
    hipStream_t st = (hipStream_t)stream;
    if (rc == MFSR_OK) {
        if (mode == MFSR_DIST_STRIPES) {
            rc = process_stripes(d, frames, out16, status, st);
        } else {
            rc = drain_back(d, st);  // a pipelined STRIPES burst before: its back half first
            if (rc == MFSR_OK && d->gatherPending) {  // a STRIPES burst before: its gather uses the transport on the comm stream
                hipError_t e = hipStreamWaitEvent(st, d->evGathered, 0);
                if (e != hipSuccess) rc = (int)e;
                d->gatherPending = false;
            }
            if (rc == MFSR_OK) rc = process_reduce(d, frames, mode, out16, status, st);
        }
    }
    if (rc != MFSR_OK) d->T->abort();
    return rc;
}

extern "C" int mfsr_dist_wait_output(mfsr_dist* d, mfsr_stream_t stream)
{
    D_REQUIRE(d != nullptr);
    // pipelined STRIPES bursts: the last burst's fuse / finish / gather are enqueued here -- transport calls, so every rank
    // calls this (as it calls process_burst)
    const int rc = drain_back(d, (hipStream_t)stream);
    if (rc != MFSR_OK) {
        if (d->T) d->T->abort();
        return rc;
    }
    if (d->gatherPending) D_HIP(hipStreamWaitEvent((hipStream_t)stream, d->evGathered, 0));
    return MFSR_OK;
}

extern "C" int mfsr_dist_wait_previous(mfsr_dist* d, mfsr_stream_t stream)
{
    D_REQUIRE(d != nullptr);
    if (d->gatherPending) D_HIP(hipStreamWaitEvent((hipStream_t)stream, d->evGathered, 0));
    return MFSR_OK;
}

extern "C" int mfsr_dist_measured_flow(mfsr_dist* d, float* maxAbsFlowY, mfsr_stream_t stream)
{
    D_REQUIRE(d && maxAbsFlowY);
    *maxAbsFlowY = 0.0f;
    if (!d->lastFlag) return MFSR_OK;  // no STRIPES burst has completed its back half yet
    if (d->gatherPending) D_HIP(hipStreamWaitEvent((hipStream_t)stream, d->evGathered, 0));
    int bits = 0;
    D_HIP(hipMemcpyAsync(&bits, d->lastFlag + 1, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    D_HIP(hipStreamSynchronize((hipStream_t)stream));
    memcpy(maxAbsFlowY, &bits, sizeof(float));
    return MFSR_OK;
}

// warp+fuse launch timing over BOTH burst contexts of a rank (pipelined bursts alternate between them)
extern "C" int mfsr_dist_timing(mfsr_dist* d, int enable)
{
    D_REQUIRE(d != nullptr);
    D_TRY(mfsr_burst_timing(d->burst, enable));
    if (d->burst2 != d->burst) D_TRY(mfsr_burst_timing(d->burst2, enable));
    return MFSR_OK;
}

extern "C" int mfsr_dist_timing_read(mfsr_dist* d, double* totalMs, int* launches, int* frames)
{
    D_REQUIRE(d && totalMs && launches && frames);
    D_TRY(mfsr_burst_timing_read(d->burst, totalMs, launches, frames));
    if (d->burst2 != d->burst) {
        double ms = 0;
        int l = 0, f = 0;
        D_TRY(mfsr_burst_timing_read(d->burst2, &ms, &l, &f));
        *totalMs += ms;
        *launches += l;
        *frames += f;
    }
    return MFSR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// G ranks in one process: one worker thread per rank drives that rank's mfsr_dist (LocalTransport)

struct mfsr_dist_group {
    LocalWorld world;
    int G = 0;
    std::vector<mfsr_dist*> ranks;
    std::vector<hipStream_t> own;  // per-rank compute streams, used when the caller passes none
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv;
    // one job at a time: process_burst (kind 1) or wait_output (kind 2)
    long long generation = 0;
    int kind = 0, pendingWorkers = 0;
    bool quit = false;
    const uint16_t* const* frames = nullptr;  // [G * N]
    int mode = 0;
    uint16_t* out16 = nullptr;
    int* const* status = nullptr;
    const mfsr_stream_t* streams = nullptr;
    std::vector<int> rc;
};

static void group_worker(mfsr_dist_group* g, int r)
{
    (void)hipSetDevice(g->world.dev[r]);
    long long seen = 0;
    for (;;) {
        int kind;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv.wait(lk, [&] { return g->quit || g->generation != seen; });
            if (g->quit) return;
            seen = g->generation;
            kind = g->kind;
        }
        mfsr_dist* d = g->ranks[r];
        const mfsr_stream_t st = g->streams ? g->streams[r] : (mfsr_stream_t)g->own[r];
        int rc = MFSR_OK;
        if (kind == 1) {
            rc = mfsr_dist_process_burst(d, g->frames + (size_t)r * d->cfg.frames, g->mode, r == 0 ? g->out16 : nullptr,
                                         g->status ? g->status[r] : nullptr, st);
        } else if (kind == 2) {
            rc = mfsr_dist_wait_output(d, st);
        } else if (kind == 3) {
            rc = mfsr_dist_wait_output(d, st);
            hipError_t e = hipStreamSynchronize((hipStream_t)st);
            if (e == hipSuccess) e = hipStreamSynchronize(d->commStream);
            if (rc == MFSR_OK && e != hipSuccess) rc = (int)e;
        }
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->rc[r] = rc;
            g->pendingWorkers--;
        }
        g->cv.notify_all();
    }
}

static int group_run(mfsr_dist_group* g, int kind)
{
    std::unique_lock<std::mutex> lk(g->mu);
    g->kind = kind;
    g->pendingWorkers = g->G;
    g->generation++;
    g->cv.notify_all();
    g->cv.wait(lk, [&] { return g->pendingWorkers == 0; });
    for (int r = 0; r < g->G; r++)
        if (g->rc[r] != MFSR_OK) return g->rc[r];
    return MFSR_OK;
}

extern "C" void mfsr_dist_group_destroy(mfsr_dist_group* g)
{
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->quit = true;
    }
    g->cv.notify_all();
    for (auto& t : g->workers)
        if (t.joinable()) t.join();
    int cur = 0;
    const bool haveCur = hipGetDevice(&cur) == hipSuccess;
    for (int r = 0; r < (int)g->ranks.size(); r++) {
        (void)hipSetDevice(g->world.dev[r]);
        if (g->own[r]) {
            (void)hipStreamSynchronize(g->own[r]);
            (void)hipStreamDestroy(g->own[r]);
        }
        if (g->ranks[r]) dist_free(g->ranks[r]);
    }
    if (haveCur) (void)hipSetDevice(cur);
    delete g;
}

extern "C" int mfsr_dist_group_create(mfsr_dist_group** out, const mfsr_config* cfg, int worldSize, const int* devices,
                                      void* const* workspaces, size_t workspaceBytes)
{
    D_REQUIRE(out && cfg && worldSize >= 1 && worldSize <= 64 && workspaces);
    int nDev = 0, cur = 0;
    D_HIP(hipGetDeviceCount(&nDev));
    D_HIP(hipGetDevice(&cur));
    for (int r = 0; r < worldSize; r++) {
        D_REQUIRE(workspaces[r] != nullptr);
        D_REQUIRE(!devices || (devices[r] >= 0 && devices[r] < nDev));
    }
    mfsr_dist_group* g = new (std::nothrow) mfsr_dist_group;
    D_REQUIRE(g != nullptr);
    g->G = worldSize;
    g->world.G = worldSize;
    g->world.dev.resize(worldSize);
    for (int r = 0; r < worldSize; r++) g->world.dev[r] = devices ? devices[r] : cur;
    g->world.box.resize((size_t)worldSize * worldSize);
    if (const char* e = getenv("MFSR_DIST_TIMEOUT_S")) {
        const double t = atof(e);
        if (t > 0) g->world.timeoutSec = t;
    }
    g->ranks.assign(worldSize, nullptr);
    g->own.assign(worldSize, nullptr);
    g->rc.assign(worldSize, MFSR_OK);
    int rc = MFSR_OK;
    // peer access between distinct devices (a copy still works without it, staged by the runtime)
    for (int a = 0; a < worldSize && rc == MFSR_OK; a++) {
        for (int b = 0; b < worldSize; b++) {
            const int da = g->world.dev[a], db = g->world.dev[b];
            if (da == db) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can) continue;
            if (hipSetDevice(da) != hipSuccess) continue;
            (void)hipDeviceEnablePeerAccess(db, 0);  // "already enabled" is fine
            (void)hipGetLastError();
        }
    }
    for (int r = 0; r < worldSize && rc == MFSR_OK; r++) {
        hipError_t he = hipSetDevice(g->world.dev[r]);
        if (he != hipSuccess) {
            rc = (int)he;
            break;
        }
        he = hipStreamCreateWithFlags(&g->own[r], hipStreamNonBlocking);
        if (he != hipSuccess) {
            rc = (int)he;
            break;
        }
        rc = dist_new(&g->ranks[r], cfg, r, worldSize, workspaces[r], workspaceBytes);
        if (rc != MFSR_OK) break;
        LocalTransport* t = new (std::nothrow) LocalTransport;
        rc = t ? t->init(&g->world, r, g->world.dev[r], g->ranks[r]->flag + 8) : MFSR_E_INVALID;
        if (rc != MFSR_OK) {
            delete t;
            break;
        }
        g->ranks[r]->T = t;
    }
    (void)hipSetDevice(cur);
    if (rc != MFSR_OK) {
        mfsr_dist_group_destroy(g);
        return rc;
    }
    for (int r = 0; r < worldSize; r++) g->workers.emplace_back(group_worker, g, r);
    *out = g;
    return MFSR_OK;
}

extern "C" mfsr_dist* mfsr_dist_group_rank(mfsr_dist_group* g, int rank)
{
    return (g && rank >= 0 && rank < g->G) ? g->ranks[rank] : nullptr;
}

extern "C" int mfsr_dist_group_set_raw_halo(mfsr_dist_group* g, int rawHalo)
{
    D_REQUIRE(g != nullptr);
    for (int r = 0; r < g->G; r++) D_TRY(mfsr_dist_set_raw_halo(g->ranks[r], rawHalo));
    return MFSR_OK;
}

extern "C" int mfsr_dist_group_process_burst(mfsr_dist_group* g, const uint16_t* const* frames, int mode, uint16_t* out16,
                                             int* const* status, const mfsr_stream_t* streams)
{
    D_REQUIRE(g && frames && out16);
    g->frames = frames;
    g->mode = mode;
    g->out16 = out16;
    g->status = status;
    g->streams = streams;
    return group_run(g, 1);
}

extern "C" int mfsr_dist_group_wait_output(mfsr_dist_group* g, const mfsr_stream_t* streams)
{
    D_REQUIRE(g != nullptr);
    g->streams = streams;
    return group_run(g, 2);
}

extern "C" int mfsr_dist_group_synchronize(mfsr_dist_group* g, const mfsr_stream_t* streams)
{
    D_REQUIRE(g != nullptr);
    g->streams = streams;
    return group_run(g, 3);
}
