// pipeline.cpp -- burst driver: N raw frames in -> one x-s frame out.
//
// This is the L3 layer the reference lacks for its ImageStackAlignator kernels
// (the order below is reconstructed from the kernels'
// data dependencies; the reference's only working burst driver,
// finalProject/Project/multi_frame_sr.cpp:146-209, drives third-party OpenCV
// code).  Stage letters follow SURVEY.md section 3.3:
//
//   set_reference : A1 half-res RGB, tracking pyramid (gray -> gaussin_filter_1D
//                   prefilter -> 2x box levels), E structure tensor -> smooth ->
//                   ComputeKernelParam, A2+A3 debayered fallback image
//   add_frame     : A1, tracking pyramid, B tile tracking coarse -> fine,
//                   D CreateFlowFieldFromTiles + K Lucas-Kanade iterations,
//                   F ComputeRobustnessMask, G accumulate onto the HR grid
//   finish        : H ApplyWeighting (+fallback) + GammasRGB (+quantise)
//
// The joint shift minimiser (C) is the identity when only (reference, k) pairs
// are measured, which is the per-frame streaming / frame-sharded mode this
// driver implements; it is exposed separately (mfsr_minimizeShifts).
//
// Everything runs on the caller's stream out of the caller's workspace: no
// allocation, no synchronisation, so a whole add_frame chain can be captured in
// a hipGraph.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "common.hpp"

namespace {

constexpr int kMaxLevels = 4;
constexpr int kMaxTimedLaunches = 4096;
// Per-frame products (flow, certainty mask) live in a ring of kRing slots: a frame's slot stays
// untouched until the warp+fuse that consumes it has run, which with cfg.asyncFuse happens on the
// burst's own stream while the caller's stream already aligns the next frames.
constexpr int kRing = 2 * MFSR_MAX_FUSE_GROUP;  // a group waiting to be fused + the group the fuse stream is reading
constexpr int kMaxUploadRing = 32;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Img {
    void* ptr = nullptr;
    int pitch = 0;
    int w = 0, h = 0;
};

struct Layout {
    // geometry
    int W, H, hw, hh, tw, th, hrW, hrH;
    int flowScale;
    int lw[kMaxLevels], lh[kMaxLevels];      // level image dims
    int tcx[kMaxLevels], tcy[kMaxLevels];    // tile counts per level
    int maxFactor;
    // buffers
    Img refHalf, movHalf;                    // float3, half res
    Img refPyr[8], movPyr[8];                // float, factor 1,2,4,.. (index = log2 factor)
    Img kparam4;                             // float4, tracking res
    Img tensor, tensorTmp, tensorSm;         // float3, tracking res
    Img fallback;                            // float3, W x H
    Img tmpA, tmpB;                          // float, tracking res
    Img flowBuf[2 * kRing];                  // float2, tracking res: per ring slot the two buffers of the LK ping-pong
    Img maskBuf[kRing];                      // float4, half res, one per ring slot
    Img shifts[kMaxLevels], pre[kMaxLevels]; // float2, tile grids
    // unfused path scratch
    Img warped, Ix, Iy, It, rawf;
    Img lkSum[2], lkDiff[2];                 // fused LK: (warped + ref) / (warped - ref) handed from iteration to iteration
    float *refTiles, *movTiles, *cc, *boxX, *boxY, *sqsum, *dist;
    float* refSq[kMaxLevels];  // fused tracker: sum(ref^2) per tile and level, taken once per reference
    // global pre-alignment (cfg.preAlign): search pyramids of the reference and the moved frame, workspace, result
    void *preRefPyr, *preMovPyr, *preWork;
    mfsr_prealign* preResult;
    // frame-batched alignment (align_group): the moved frame's intermediates of frames 1 .. group-1 of a group (frame 0
    // uses the members above); swapped in around the per-frame launches
    struct AlignSet {
        Img movHalf, movPyr[8], shifts[kMaxLevels], lkSum[2], lkDiff[2];
        void* preMovPyr;
        mfsr_prealign* preResult;
    } sets[MFSR_MAX_FUSE_GROUP - 1];
    int nSets;
    // host-frame bursts (cfg.uploadRing): device slots the library uploads into
    uint16_t* rawRing[kMaxUploadRing];
    uint16_t* refRaw[2];
    size_t total;
};

struct Bump {
    char* base;
    size_t off;
    void* take(size_t bytes)
    {
        off = align_up(off, 256);
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
    Img image(int w, int h, int elemBytes)
    {
        Img im;
        im.w = w;
        im.h = h;
        im.pitch = (int)align_up((size_t)w * elemBytes, 64);
        im.ptr = take((size_t)im.pitch * h);
        return im;
    }
};

int ilog2(int v)
{
    int l = 0;
    while ((1 << l) < v) l++;
    return l;
}

int validate(const mfsr_config* c)
{
    MFSR_REQUIRE(c != nullptr);
    MFSR_REQUIRE(c->width >= 64 && c->height >= 64 && (c->width % 4) == 0 && (c->height % 4) == 0);
    MFSR_REQUIRE(c->frames >= 1 && c->reference >= 0 && c->reference < c->frames);
    MFSR_REQUIRE(c->scale >= 1 && c->scale <= 8);
    MFSR_REQUIRE(c->levels >= 1 && c->levels <= kMaxLevels);
    MFSR_REQUIRE(c->levelFactor[c->levels - 1] == 1);
    for (int l = 0; l < c->levels; l++) {
        const int f = c->levelFactor[l];
        MFSR_REQUIRE(f >= 1 && f <= 128 && (f & (f - 1)) == 0);
        if (l > 0) MFSR_REQUIRE(c->levelFactor[l] < c->levelFactor[l - 1]);
        MFSR_REQUIRE(c->tileSize[l] >= 4 && c->tileSize[l] <= 128 && c->maxShift[l] >= 1 && c->maxShift[l] <= 15);
        MFSR_REQUIRE(c->tileSize[l] > 2 * c->maxShift[l]);
        if (c->fused) {
            // LDS need of one tile in mfsr_trackTilesFused (template + patch + row sums + distance image): fail at
            // create, not at the first add_frame (tileSize >~ 90); cfg.fused = 0 serves larger tiles
            const long long T = c->tileSize[l], S = c->maxShift[l], Lt = T + 2 * S, R = 2 * S + 1;
            if (4 * (T * T + Lt * (Lt + 1) + 1 + Lt * R + R * R + 4) > 64 * 1024) {
                fprintf(stderr, "mfsr: tileSize %d / maxShift %d exceeds the fused tile tracker's LDS budget (use cfg.fused = 0)\n",
                        (int)T, (int)S);
                return MFSR_E_UNSUPPORTED;
            }
        }
    }
    MFSR_REQUIRE(c->lkIterations >= 0 && c->lkHalfWindow >= 0 && c->lkHalfWindow <= 15);
    MFSR_REQUIRE(c->maxVal > 0);
    if (c->preAlign) MFSR_REQUIRE(c->preAlignMaxAngle >= 0.0f && c->preAlignMaxAngle <= 45.0f);
    MFSR_REQUIRE(c->uploadRing == 0 || (c->uploadRing >= 3 && c->uploadRing <= kMaxUploadRing));
    MFSR_REQUIRE(c->pairFrames >= 0 && c->pairFrames <= MFSR_MAX_FUSE_GROUP);
    return MFSR_OK;
}

void make_layout(const mfsr_config* c, char* base, Layout* L)
{
    memset((void*)L, 0, sizeof(*L));
    Bump b{base, 0};
    L->W = c->width;
    L->H = c->height;
    L->hw = c->width / 2;
    L->hh = c->height / 2;
    L->tw = c->mono ? L->W : L->hw;
    L->th = c->mono ? L->H : L->hh;
    L->flowScale = c->mono ? 1 : 2;
    L->hrW = c->width * c->scale;
    L->hrH = c->height * c->scale;
    L->maxFactor = c->levelFactor[0];
    const int nl = ilog2(L->maxFactor) + 1;

    L->refHalf = b.image(L->hw, L->hh, 12);
    L->movHalf = b.image(L->hw, L->hh, 12);
    for (int i = 0; i < nl; i++) {
        L->refPyr[i] = b.image(L->tw >> i, L->th >> i, 4);
        L->movPyr[i] = b.image(L->tw >> i, L->th >> i, 4);
    }
    L->kparam4 = b.image(L->tw, L->th, 16);
    L->tensor = b.image(L->tw, L->th, 12);
    L->tensorTmp = b.image(L->tw, L->th, 12);
    L->tensorSm = b.image(L->tw, L->th, 12);
    L->fallback = b.image(L->W, L->H, 12);
    L->tmpA = b.image(L->tw, L->th, 4);
    L->tmpB = b.image(L->tw, L->th, 4);
    for (int i = 0; i < 2 * kRing; i++) L->flowBuf[i] = b.image(L->tw, L->th, 8);
    for (int i = 0; i < kRing; i++) L->maskBuf[i] = b.image(L->hw, L->hh, 16);
    size_t maxTileFloats = 0, maxTiles = 0, maxDist = 0;
    for (int l = 0; l < c->levels; l++) {
        const int f = c->levelFactor[l];
        L->lw[l] = L->tw / f;
        L->lh[l] = L->th / f;
        L->tcx[l] = L->lw[l] / c->tileSize[l] > 0 ? L->lw[l] / c->tileSize[l] : 1;
        L->tcy[l] = L->lh[l] / c->tileSize[l] > 0 ? L->lh[l] / c->tileSize[l] : 1;
        L->shifts[l] = b.image(L->tcx[l], L->tcy[l], 8);
        L->pre[l] = b.image(L->tcx[l], L->tcy[l], 8);
        const size_t tiles = (size_t)L->tcx[l] * L->tcy[l];
        const size_t Lt = (size_t)c->tileSize[l] + 2 * c->maxShift[l];
        const size_t R = 2 * (size_t)c->maxShift[l] + 1;
        if (tiles * Lt * Lt > maxTileFloats) maxTileFloats = tiles * Lt * Lt;
        if (tiles > maxTiles) maxTiles = tiles;
        if (tiles * R * R > maxDist) maxDist = tiles * R * R;
    }
    for (int l = 0; l < kMaxLevels; l++) L->refSq[l] = nullptr;
    if (c->fused) {
        for (int l = 0; l < c->levels; l++) L->refSq[l] = (float*)b.take((size_t)L->tcx[l] * L->tcy[l] * 4);
        for (int i = 0; i < 2; i++) {
            L->lkSum[i] = b.image(L->tw, L->th, 4);
            L->lkDiff[i] = b.image(L->tw, L->th, 4);
        }
    }
    L->Ix = b.image(L->tw, L->th, 4);  // (fused: the derivative images of the kernel-parameter chain, made on the fuse stream)
    L->Iy = b.image(L->tw, L->th, 4);
    if (!c->fused) {
        L->warped = b.image(L->tw, L->th, 4);
        L->It = b.image(L->tw, L->th, 4);
        L->rawf = b.image(L->W, L->H, 4);
        L->refTiles = (float*)b.take(maxTileFloats * 4);
        L->movTiles = (float*)b.take(maxTileFloats * 4);
        L->cc = (float*)b.take(maxTileFloats * 4);
        L->boxX = (float*)b.take(maxTileFloats * 4);
        L->boxY = (float*)b.take(maxTileFloats * 4);
        L->sqsum = (float*)b.take(maxTiles * 4);
        L->dist = (float*)b.take(maxDist * 4);
    }
    if (c->preAlign) {
        const size_t pb = mfsr_preAlign_pyramid_bytes(L->tw, L->th);
        L->preRefPyr = b.take(pb);
        L->preMovPyr = b.take(pb);
        L->preWork = b.take(mfsr_preAlign_workspace_bytes(c->preAlignMaxAngle));
        L->preResult = (mfsr_prealign*)b.take(sizeof(mfsr_prealign));
    }
    L->nSets = 0;
    if (c->fused && mfsr_burst_group_size(c) > 1) {
        L->nSets = mfsr_burst_group_size(c) - 1;
        for (int q = 0; q < L->nSets; q++) {
            Layout::AlignSet& S = L->sets[q];
            S.movHalf = b.image(L->hw, L->hh, 12);
            for (int i = 0; i < nl; i++) S.movPyr[i] = b.image(L->tw >> i, L->th >> i, 4);
            for (int l = 0; l < c->levels; l++) S.shifts[l] = b.image(L->tcx[l], L->tcy[l], 8);
            for (int i = 0; i < 2; i++) {
                S.lkSum[i] = b.image(L->tw, L->th, 4);
                S.lkDiff[i] = b.image(L->tw, L->th, 4);
            }
            S.preMovPyr = nullptr;
            S.preResult = nullptr;
            if (c->preAlign) {
                S.preMovPyr = b.take(mfsr_preAlign_pyramid_bytes(L->tw, L->th));
                S.preResult = (mfsr_prealign*)b.take(sizeof(mfsr_prealign));
            }
        }
    }
    for (int i = 0; i < c->uploadRing; i++) L->rawRing[i] = (uint16_t*)b.take((size_t)L->W * L->H * 2);
    if (c->uploadRing > 0)
        for (int i = 0; i < 2; i++) L->refRaw[i] = (uint16_t*)b.take((size_t)L->W * L->H * 2);
    L->total = align_up(b.off, 256);
}

// exchange the moved-frame intermediates named by the Layout members with those of an align set
void swap_set(Layout& L, Layout::AlignSet& S)
{
    std::swap(L.movHalf, S.movHalf);
    for (int i = 0; i < 8; i++) std::swap(L.movPyr[i], S.movPyr[i]);
    for (int l = 0; l < kMaxLevels; l++) std::swap(L.shifts[l], S.shifts[l]);
    for (int i = 0; i < 2; i++) {
        std::swap(L.lkSum[i], S.lkSum[i]);
        std::swap(L.lkDiff[i], S.lkDiff[i]);
    }
    std::swap(L.preMovPyr, S.preMovPyr);
    std::swap(L.preResult, S.preResult);
}

mfsr_tex2d as_tex(const Img& im)
{
    mfsr_tex2d t;
    t.ptr = im.ptr;
    t.pitch = im.pitch;
    t.width = im.w;
    t.height = im.h;
    return t;
}

}  // namespace

struct mfsr_burst {
    mfsr_config cfg;
    Layout L;
    float taps[99];
    int ntaps;
    float tensorTaps[99];
    int ntensorTaps;
    Img* flowCur;  // flow of the last add_frame (raw-pixel units)
    Img* maskCur;  // certainty mask of the last add_frame
    mfsr_prealign* preCur;  // pre-alignment estimate of the last aligned frame (cfg.preAlign; device memory)
    bool haveRef;
    bool refStale;                      // the reference's alignment products were swapped away (process_joint): finish is
                                        // still valid, aligning another frame needs a new set_reference
    // frame grouping (cfg.pairFrames): aligned frames wait here until their group is complete, then the
    // group is fused in one pass over the accumulators; flush/finish fuses what is left of a group
    struct Pending {
        int n;  // frames waiting (0 .. group - 1)
        int slot[MFSR_MAX_FUSE_GROUP];
        const uint16_t* raw[MFSR_MAX_FUSE_GROUP];
        Img* flow[MFSR_MAX_FUSE_GROUP];
        Img* mask[MFSR_MAX_FUSE_GROUP];
        bool deferred[MFSR_MAX_FUSE_GROUP];  // not aligned yet: align_deferred aligns the waiting frames as one batch
        int isRef[MFSR_MAX_FUSE_GROUP];
        mfsr_float3 *imgOut, *totalWeights;
    } pend;
    // host bursts: the burst's second-to-last group, aligned and waiting with the last one (pend) for mfsr_burst_finish_host,
    // which fuses both band by band (heldHas: the entries of `held` are valid)
    Pending held;
    bool heldHas;
    int group;  // frames per warp+fuse launch (mfsr_burst_group_size)
    int nFramesTimed;
    // mfsr_burst_begin: these accumulators are to be overwritten by the first fuse instead of zeroed
    struct Fresh {
        bool has;
        mfsr_float3 *imgOut, *totalWeights;
    } fresh;
    // ring + asynchronous fuse (cfg.asyncFuse)
    int frameCounter;
    hipStream_t fuseStream;             // high priority, non-blocking; null without asyncFuse
    hipEvent_t evAligned[kRing];        // recorded on the caller's stream when slot's flow/mask are complete
    hipEvent_t evFused[kRing];          // recorded on fuseStream when the fuse that read the slot is done
    // the reference products only the fuse and the finish read (kernel parameters, debayered fallback image) are made on
    // fuseStream, beside the alignment of the first group on the caller's stream: off the critical path of the burst
    hipEvent_t evRefStart, evRefDone;
    bool refOnFuse;                     // evRefDone is pending for consumers on other streams than fuseStream
    bool fusedOutstanding[kRing];       // evFused[slot] recorded and not yet waited for by the caller's stream
    Img* slotFlow[kRing];               // which buffer of the slot's LK ping-pong pair holds the frame's final flow
    // host-frame bursts (cfg.uploadRing): copy stream, per-slot events, reference double buffer
    hipStream_t copyStream;
    hipStream_t downStream;                 // D2H of the finished image (mfsr_burst_finish_host), concurrent with the uploads
    hipEvent_t evFinished, evDown;          // finish kernel done (compute stream) / D2H done (down stream)
    bool downRecorded;
    hipEvent_t evBand[16];                  // host bursts: band i of the output image finished (mfsr_burst_finish_host)
    int framesSinceRef;                     // frames added since the last set_reference
    bool holdLastGroup;                     // host bursts: the group the burst's last frame completes waits for finish_host
    hipEvent_t evUp[kMaxUploadRing + 2];    // upload of the slot complete (copy stream); [ring..ring+1] = reference slots
    hipEvent_t evFree[kMaxUploadRing + 2];  // last consumer of the slot enqueued (compute / fuse stream)
    bool freeRecorded[kMaxUploadRing + 2];
    // mfsr_burst_prefetch_host: frames whose upload is already enqueued, in order; consumed by add_frame_host
    struct Prefetched {
        const uint16_t* host;
        int slot;
    } prefetched[kMaxUploadRing];
    int nPrefetched, prefetchHead;
    bool upPending;                         // uploads were enqueued that the compute stream has not been made to wait for yet
    int upPendingSlot;                      // ... the LAST of them (the copy stream is in order: its event covers the earlier ones)
    int refSlot;                            // upload slot of the current host reference (-1: none / released)
    bool hostBusy;                          // this host burst was enqueued before the previous one's download had finished
    int upCounter, refCounter;
    const uint16_t* refHost;                // host pointer of the current reference and its device copy
    const uint16_t* refDev;
    // mfsr_stream: the per-frame products (half-res RGB, tracking pyramid, search pyramid) of the next set_reference /
    // add_frame are already in L.refHalf.. / L.movHalf.. (the stream keeps them per frame and swaps the descriptors in)
    bool refPrepared, movPrepared;
    // joint mode (stage C): tile shifts of the frame being aligned come from the minimiser instead of the tracker
    const Img* givenShifts;
    std::vector<float> jointHost;  // host staging of the joint mode's design matrix / pointer table (must outlive the copies)
    // optional per-launch timing of the accumulate kernel (bench.py roofline leg)
    bool timing;
    int nEvents;
    hipEvent_t evStart[kMaxTimedLaunches], evStop[kMaxTimedLaunches];
};

#define TRY(expr)                  \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != MFSR_OK) return rc_; \
    } while (0)

static int flush_pending(mfsr_burst* b, mfsr_stream_t stream, bool materializeFresh = true);


extern "C" int mfsr_config_default(mfsr_config* cfg, int width, int height, int frames, int scale, int mono)
{
    MFSR_REQUIRE(cfg != nullptr);
    memset(cfg, 0, sizeof(*cfg));
    cfg->width = width;
    cfg->height = height;
    cfg->frames = frames;
    cfg->reference = 0;
    cfg->scale = scale;
    cfg->mono = mono;
    if (mono) {
        cfg->cfa[0] = cfg->cfa[1] = cfg->cfa[2] = cfg->cfa[3] = MFSR_GREEN;
    } else {
        cfg->cfa[0] = MFSR_RED;
        cfg->cfa[1] = MFSR_GREEN;
        cfg->cfa[2] = MFSR_GREEN;
        cfg->cfa[3] = MFSR_BLUE;
    }
    // 12-bit sensor, black level 256 (SURVEY.md section 8d synthetic-input recipe)
    for (int i = 0; i < 3; i++) {
        cfg->black[i] = 256.0f;
        cfg->white[i] = 4095.0f - 256.0f;
    }
    cfg->maxVal = 4095.0f;
    cfg->levels = 2;
    cfg->levelFactor[0] = 2;
    cfg->levelFactor[1] = 1;
    cfg->tileSize[0] = 32;
    cfg->tileSize[1] = 32;
    cfg->maxShift[0] = 4;
    cfg->maxShift[1] = 4;
    cfg->minimumThreshold = 0.0f;
    cfg->sigmaTracking = 0.5f;  // SigmaDebayerTracking, test_opencv/main.cpp:1868
    cfg->lkIterations = 3;
    cfg->lkHalfWindow = 3;
    cfg->lkMinDet = 1e-4f;
    cfg->alpha = 1e-4f;
    cfg->beta = 1e-6f;
    cfg->thresholdM = 1.0f;
    cfg->sigmaTensor = 1.0f;
    cfg->Dth = 0.005f;
    cfg->Dtr = 0.05f;
    cfg->kDetail = 0.3f;
    cfg->kDenoise = 2.0f;
    cfg->kStretch = 2.0f;
    cfg->kShrink = 2.0f;
    cfg->weightThreshold = 1e-3f;
    cfg->applyGamma = 0;
    cfg->fused = 1;
    cfg->pairFrames = 1;
    cfg->preAlign = 0;   // opt-in: bursts with rotations / shifts beyond the tile tracker's reach (the bundled "city" burst)
    cfg->preAlignMaxAngle = 20.0f;
    cfg->asyncFuse = 0;  // 1: warp+fuse launches on a burst-owned stream beside the alignment of the next group.  It paid +5 % in
                         // round 2; since the fuse kernel keeps four 128-VGPR workgroups per CU (round 3) nothing of the alignment
                         // gets onto a CU beside it -- the two streams alternate -- and the fork / join events cost more than
                         // the overlap returns: 5.97 ms per 4K x 16 burst on one stream against 6.03 on two (interleaved A/B,
                         // profiles/r04_async_fuse_ab.txt).  Off by default since round 4; bit-identical either way.
    return MFSR_OK;
}

extern "C" size_t mfsr_burst_workspace_bytes(const mfsr_config* cfg)
{
    if (validate(cfg) != MFSR_OK) return 0;
    Layout L;
    make_layout(cfg, nullptr, &L);
    return L.total;
}

extern "C" size_t mfsr_burst_accumulator_bytes(const mfsr_config* cfg)
{
    if (!cfg || cfg->width <= 0 || cfg->height <= 0 || cfg->scale <= 0) return 0;
    return (size_t)12 * cfg->scale * cfg->width * (size_t)cfg->scale * cfg->height;
}

extern "C" int mfsr_burst_create(mfsr_burst** out, const mfsr_config* cfg, void* workspace, size_t workspaceBytes)
{
    MFSR_REQUIRE(out != nullptr && workspace != nullptr);
    TRY(validate(cfg));
    if (mfsr_device_count() <= 0) {
        fprintf(stderr, "mfsr: no HIP device available (there is no CPU fallback)\n");
        return MFSR_E_NODEVICE;
    }
    MFSR_REQUIRE(((uintptr_t)workspace & 255) == 0);
    mfsr_burst* b = new (std::nothrow) mfsr_burst;
    MFSR_REQUIRE(b != nullptr);
    b->cfg = *cfg;
    make_layout(cfg, (char*)workspace, &b->L);
    if (b->L.total > workspaceBytes) {
        fprintf(stderr, "mfsr: workspace too small: need %zu bytes, got %zu\n", b->L.total, workspaceBytes);
        delete b;
        return MFSR_E_WORKSPACE;
    }
    b->ntaps = mfsr_gaussin_filter_1D(cfg->sigmaTracking, b->taps);
    b->ntensorTaps = mfsr_gaussin_filter_1D(cfg->sigmaTensor, b->tensorTaps);
    b->flowCur = &b->L.flowBuf[0];
    b->maskCur = &b->L.maskBuf[0];
    b->preCur = b->L.preResult;
    b->pend.n = 0;
    b->group = mfsr_burst_group_size(&b->cfg);
    b->fresh.has = false;
    b->nFramesTimed = 0;
    b->timing = false;
    b->nEvents = 0;
    memset(b->evStart, 0, sizeof(b->evStart));
    memset(b->evStop, 0, sizeof(b->evStop));
    b->frameCounter = 0;
    b->fuseStream = nullptr;
    b->evRefStart = b->evRefDone = nullptr;
    b->refOnFuse = false;
    for (int i = 0; i < kRing; i++) {
        b->evAligned[i] = b->evFused[i] = nullptr;
        b->fusedOutstanding[i] = false;
        b->slotFlow[i] = nullptr;
    }
    if (cfg->asyncFuse) {
        int lo = 0, hi = 0;
        hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        // MFSR_FUSE_PRIO=low|normal (A/B): the fuse stream's priority against the caller's (default: high)
        static const int prioMode = [] {
            const char* v = getenv("MFSR_FUSE_PRIO");
            return !v ? 0 : (v[0] == 'l' ? 1 : (v[0] == 'n' ? 2 : 0));
        }();
        const int prio = prioMode == 1 ? lo : (prioMode == 2 ? (lo + hi) / 2 : hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&b->fuseStream, hipStreamNonBlocking, prio);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->evRefStart, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->evRefDone, hipEventDisableTiming);
        for (int i = 0; i < kRing && e == hipSuccess; i++) {
            e = hipEventCreateWithFlags(&b->evAligned[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&b->evFused[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            mfsr_burst_destroy(b);
            return MFSR_E_NODEVICE;
        }
    }
    b->refPrepared = b->movPrepared = false;
    b->givenShifts = nullptr;
    b->copyStream = nullptr;
    b->downStream = nullptr;
    b->evFinished = b->evDown = nullptr;
    b->downRecorded = false;
    b->upCounter = b->refCounter = 0;
    for (int i = 0; i < 16; i++) b->evBand[i] = nullptr;
    b->framesSinceRef = 0;
    b->holdLastGroup = false;
    b->heldHas = false;
    b->held.n = 0;
    b->refHost = b->refDev = nullptr;
    b->refSlot = -1;
    b->hostBusy = false;
    b->upPending = false;
    b->upPendingSlot = -1;
    b->nPrefetched = b->prefetchHead = 0;
    for (int i = 0; i < kMaxUploadRing + 2; i++) {
        b->evUp[i] = b->evFree[i] = nullptr;
        b->freeRecorded[i] = false;
    }
    if (cfg->uploadRing > 0) {
        hipError_t e = hipStreamCreateWithFlags(&b->copyStream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->downStream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->evFinished, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->evDown, hipEventDisableTiming);
        for (int i = 0; i < cfg->uploadRing + 2 && e == hipSuccess; i++) {
            e = hipEventCreateWithFlags(&b->evUp[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&b->evFree[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            mfsr_burst_destroy(b);
            return MFSR_E_NODEVICE;
        }
    }
    b->haveRef = false;
    b->refStale = false;
    *out = b;
    return MFSR_OK;
}

extern "C" void mfsr_burst_destroy(mfsr_burst* b)
{
    if (!b) return;
    for (int i = 0; i < kMaxTimedLaunches; i++) {
        if (b->evStart[i]) (void)hipEventDestroy(b->evStart[i]);
        if (b->evStop[i]) (void)hipEventDestroy(b->evStop[i]);
    }
    if (b->fuseStream) (void)hipStreamSynchronize(b->fuseStream);
    for (int i = 0; i < kRing; i++) {
        if (b->evAligned[i]) (void)hipEventDestroy(b->evAligned[i]);
        if (b->evFused[i]) (void)hipEventDestroy(b->evFused[i]);
    }
    if (b->evRefStart) (void)hipEventDestroy(b->evRefStart);
    if (b->evRefDone) (void)hipEventDestroy(b->evRefDone);
    if (b->fuseStream) (void)hipStreamDestroy(b->fuseStream);
    if (b->copyStream) (void)hipStreamSynchronize(b->copyStream);
    if (b->downStream) (void)hipStreamSynchronize(b->downStream);
    if (b->evFinished) (void)hipEventDestroy(b->evFinished);
    if (b->evDown) (void)hipEventDestroy(b->evDown);
    for (int i = 0; i < 16; i++)
        if (b->evBand[i]) (void)hipEventDestroy(b->evBand[i]);
    if (b->downStream) (void)hipStreamDestroy(b->downStream);
    for (int i = 0; i < kMaxUploadRing + 2; i++) {
        if (b->evUp[i]) (void)hipEventDestroy(b->evUp[i]);
        if (b->evFree[i]) (void)hipEventDestroy(b->evFree[i]);
    }
    if (b->copyStream) (void)hipStreamDestroy(b->copyStream);
    delete b;
}

// Per-launch timing of the warp+fuse (accumulate) kernel with HIP events recorded
// on the caller's stream around each launch.  enable != 0 starts a fresh series.
extern "C" int mfsr_burst_timing(mfsr_burst* b, int enable)
{
    MFSR_REQUIRE(b != nullptr);
    b->timing = enable != 0;
    b->nEvents = 0;
    b->nFramesTimed = 0;
    return MFSR_OK;
}

// Synchronises with the recorded events and returns the summed kernel time and
// the number of timed launches since mfsr_burst_timing(b, 1).
extern "C" int mfsr_burst_timing_read(mfsr_burst* b, double* totalMs, int* launches, int* frames)
{
    MFSR_REQUIRE(b && totalMs && launches && frames);
    double total = 0;
    for (int i = 0; i < b->nEvents; i++) {
        MFSR_HIP_TRY(hipEventSynchronize(b->evStop[i]));
        float ms = 0;
        MFSR_HIP_TRY(hipEventElapsedTime(&ms, b->evStart[i], b->evStop[i]));
        total += ms;
    }
    *totalMs = total;
    *launches = b->nEvents;
    *frames = b->nFramesTimed;
    b->nEvents = 0;
    b->nFramesTimed = 0;
    return MFSR_OK;
}

// host-frame bursts: `stream` is about to read raw frames the copy stream uploads -- wait for the last upload enqueued so far
static int wait_uploads(mfsr_burst* b, mfsr_stream_t stream)
{
    if (b->upPending) {
        MFSR_HIP_TRY(hipStreamWaitEvent(mfsr_s(stream), b->evUp[b->upPendingSlot], 0));
        b->upPending = false;
    }
    return MFSR_OK;
}

// A1 + tracking pyramid for one frame (shared by reference and moved frames)
static int prepare_frame(mfsr_burst* b, const uint16_t* raw, Img& half, Img* pyr, mfsr_stream_t stream)
{
    const mfsr_config& c = b->cfg;
    Layout& L = b->L;
    TRY(mfsr_set_cfa_pattern(c.cfa));
    const float maxValEff = c.mono ? 2.0f * c.maxVal : c.maxVal;  // mono: 4 "greens" x 0.5 -> mean of the quad
    const int nlev = ilog2(L.maxFactor) + 1;
    if (c.fused && !c.mono && b->ntaps / 2 <= 8 && L.tw == L.hw && L.th == L.hh) {
        // A1 + luma + prefilter + first pyramid level in one launch (bit-identical to the chain below)
        TRY(mfsr_prepareFrameFused(raw, (mfsr_float3*)half.ptr, half.pitch, maxValEff, L.hw, L.hh, (float*)pyr[0].ptr,
                                   pyr[0].pitch, nlev > 1 ? (float*)pyr[1].ptr : nullptr, nlev > 1 ? pyr[1].pitch : 0,
                                   b->taps, b->ntaps, stream));
        for (int i = 2; i < nlev; i++)
            TRY(mfsr_downsample2x((const float*)pyr[i - 1].ptr, pyr[i - 1].pitch, (float*)pyr[i].ptr, pyr[i].pitch, pyr[i].w,
                                  pyr[i].h, stream));
        return MFSR_OK;
    }
    TRY(mfsr_deBayersSubSample3(raw, (mfsr_float3*)half.ptr, maxValEff, L.hw, L.hh, half.pitch, stream));
    if (c.mono)
        TRY(mfsr_u16ToFloat(raw, (float*)L.tmpA.ptr, L.tmpA.pitch, L.W, L.H, 1.0f / c.maxVal, stream));
    else
        TRY(mfsr_rgbToGray((const mfsr_float3*)half.ptr, half.pitch, (float*)L.tmpA.ptr, L.tmpA.pitch, L.tw, L.th, stream));
    TRY(mfsr_separableFilter((const float*)L.tmpA.ptr, L.tmpA.pitch, (float*)L.tmpB.ptr, (float*)pyr[0].ptr, pyr[0].pitch,
                             L.tw, L.th, 1, b->taps, b->ntaps, stream));
    const int nl = ilog2(L.maxFactor) + 1;
    for (int i = 1; i < nl; i++)
        TRY(mfsr_downsample2x((const float*)pyr[i - 1].ptr, pyr[i - 1].pitch, (float*)pyr[i].ptr, pyr[i].pitch, pyr[i].w,
                              pyr[i].h, stream));
    return MFSR_OK;
}

// reference products.  [hrRow0, hrRow1) = the HR rows whose fuse / finish will read the kernel parameters and the fallback image
// (the whole grid for a single-GPU burst; a stripe for a rank of a multi-GPU burst: those two products are then made for
// the rows the stripe reads plus the halo their stencils need -- every kernel below works on a row window of the images,
// so the rows it makes are the bits the whole-image run makes)
static int set_reference_impl(mfsr_burst* b, const uint16_t* rawRef, int hrRow0, int hrRow1, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && rawRef);
    TRY(flush_pending(b, stream, false));  // a frame still waiting belongs to the previous reference
    const mfsr_config& c = b->cfg;
    Layout& L = b->L;
    MFSR_REQUIRE(hrRow0 >= 0 && hrRow1 > hrRow0 && hrRow1 <= L.hrH);
    const bool whole = (hrRow0 == 0 && hrRow1 == L.hrH) || !c.fused;  // (the unfused chain always makes whole images)
    TRY(wait_uploads(b, stream));
    if (!b->refPrepared) TRY(prepare_frame(b, rawRef, L.refHalf, L.refPyr, stream));
    if (c.fused)
        for (int l = 0; l < c.levels; l++) {
            const Img& ref = L.refPyr[ilog2(c.levelFactor[l])];
            TRY(mfsr_tileSquaredSums((const float*)ref.ptr, L.refSq[l], ref.w, ref.h, ref.pitch, c.maxShift[l], c.tileSize[l],
                                     L.tcx[l], L.tcy[l], stream));
        }

    if (c.preAlign) {
        TRY(mfsr_preAlign_init(L.preWork, c.preAlignMaxAngle, stream));
        if (!b->refPrepared)
            TRY(mfsr_preAlignPyramid((const float*)L.refPyr[0].ptr, L.tw, L.th, L.refPyr[0].pitch, L.preRefPyr, stream));
    }

    // What follows is read by the fuse and the finish only, not by the alignment: with cfg.asyncFuse it runs on the burst's fuse
    // stream (after everything the caller's stream has enqueued so far: the previous burst's finish reads the same images),
    // beside the alignment of the first group -- 0.2 ms off the burst's critical path, on which the GPU is otherwise
    // half empty.  MFSR_REF_OVERLAP=0: on the caller's stream (A/B).
    static const bool refOverlap = [] {
        const char* e = getenv("MFSR_REF_OVERLAP");
        return !(e && e[0] == '0');
    }();
    mfsr_stream_t es = stream;
    b->refOnFuse = false;
    if (refOverlap && b->fuseStream && c.fused) {
        MFSR_HIP_TRY(hipEventRecord(b->evRefStart, mfsr_s(stream)));
        MFSR_HIP_TRY(hipStreamWaitEvent(b->fuseStream, b->evRefStart, 0));
        es = (mfsr_stream_t)b->fuseStream;
    }
    // E: kernel shape field from the reference tracking image
    // E1 + E2 run once per burst on an LR-sized image, and E3 turns their result into the kernel ORIENTATION through an
    // eigen-decomposition that is ill-conditioned wherever the tensor is nearly isotropic (k1 != k2 even there,
    // kernel.cu:766-772): a tensor that differs in its last bits (mfsr_structureTensorFused reads the texels directly
    // instead of blending them with the ~1e-7 weights an exact-float bilinear fetch at a texel centre has) moves the tap
    // exponents by up to 0.5 there, i.e. the accumulators by 1e-3 relative -- measured, tools/parity_audit.py.  So the
    // fused pipeline also takes the bit-exact two-kernel chain here (+ ~10 us per burst).
    {
        // field rows the fuse of HR rows [hrRow0, hrRow1) samples: floor((Y + .5) * th / hrH - .5) and the next one; the
        // smoothing reads taps / 2 more on either side (clamped at the IMAGE border only: the window carries that halo)
        int f0 = 0, f1 = L.th;
        if (!whole) {
            const int per = L.hrH / L.th, halo = b->ntensorTaps / 2 + 1;
            f0 = hrRow0 / per - 1 - halo;
            f1 = (hrRow1 + per - 1) / per + 1 + halo;
            f0 = f0 < 0 ? 0 : f0;
            f1 = f1 > L.th ? L.th : f1;
        }
        const int fr = f1 - f0;
        Img& ix = L.Ix;
        Img& iy = L.Iy;
        auto rows = [&](const Img& im) { return (char*)im.ptr + (size_t)f0 * im.pitch; };
        TRY(mfsr_ComputeDerivatives2Rows(L.tw, L.th, ix.pitch, (float*)ix.ptr, (float*)iy.ptr, as_tex(L.refPyr[0]), f0, fr, es));
        TRY(mfsr_ComputeStructureTensor((const float*)rows(ix), (const float*)rows(iy), (mfsr_float3*)rows(L.tensor), L.tw, fr, ix.pitch,
                                        L.tensor.pitch, es));
        TRY(mfsr_separableFilter((const float*)rows(L.tensor), L.tensor.pitch, (float*)rows(L.tensorTmp), (float*)rows(L.tensorSm),
                                 L.tensorSm.pitch, L.tw, fr, 3, b->tensorTaps, b->ntensorTaps, es));
        TRY(mfsr_ComputeKernelParam((mfsr_float3*)rows(L.tensorSm), L.tw, fr, L.tensorSm.pitch, c.Dth, c.Dtr, c.kDetail, c.kDenoise,
                                    c.kStretch, c.kShrink, es));
        TRY(mfsr_float3ToFloat4((const mfsr_float3*)rows(L.tensorSm), L.tensorSm.pitch, (mfsr_float4*)rows(L.kparam4), L.kparam4.pitch,
                                L.tw, fr, es));
    }

    // A2 + A3: debayered reference = fallback image of ApplyWeighting (finish resamples it bilinearly: raw rows Y / s +- 1;
    // the row window starts on an even row -- the CFA phase -- and carries the 3-row stencil halo plus the 2-row ring the
    // kernels leave untouched at a window edge)
    const mfsr_float3 bp = {c.black[0], c.black[1], c.black[2]};
    const mfsr_float3 sc = {1.0f / c.white[0], 1.0f / c.white[1], 1.0f / c.white[2]};
    int r0 = 0, r1 = L.H;
    if (!whole) {
        r0 = (hrRow0 / c.scale - 8) & ~1;
        r1 = ((hrRow1 + c.scale - 1) / c.scale + 8 + 1) & ~1;
        r0 = r0 < 0 ? 0 : r0;
        r1 = r1 > L.H ? L.H : r1;
    }
    char* fb = (char*)L.fallback.ptr + (size_t)r0 * L.fallback.pitch;
    MFSR_HIP_TRY(hipMemsetAsync(fb, 0, (size_t)L.fallback.pitch * (r1 - r0), mfsr_s(es)));
    if (c.fused) {
        TRY(mfsr_deBayerFused(rawRef + (size_t)r0 * L.W, (mfsr_float3*)fb, L.fallback.pitch, L.W, r1 - r0, bp, sc, es));
        if (es != stream) {
            MFSR_HIP_TRY(hipEventRecord(b->evRefDone, b->fuseStream));
            b->refOnFuse = true;
        }
    } else {
        TRY(mfsr_u16ToFloat(rawRef, (float*)L.rawf.ptr, L.rawf.pitch, L.W, L.H, 1.0f, stream));
        TRY(mfsr_deBayerGreenKernel(L.W, L.H, (const float*)L.rawf.ptr, L.rawf.pitch, (mfsr_float3*)L.fallback.ptr,
                                    L.fallback.pitch, bp, sc, stream));
        TRY(mfsr_deBayerRedBlueKernel(L.W, L.H, (const float*)L.rawf.ptr, L.rawf.pitch, (mfsr_float3*)L.fallback.ptr,
                                      L.fallback.pitch, bp, sc, stream));
    }
    b->haveRef = true;
    b->refStale = false;
    b->framesSinceRef = 0;
    b->holdLastGroup = false;  // (set again by every mfsr_burst_add_frame_host)
    return MFSR_OK;
}

extern "C" int mfsr_burst_set_reference(mfsr_burst* b, const uint16_t* rawRef, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b != nullptr);
    return set_reference_impl(b, rawRef, 0, b->L.hrH, stream);
}

extern "C" int mfsr_burst_set_reference_rows(mfsr_burst* b, const uint16_t* rawRef, int hrRow0, int hrRow1, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b != nullptr);
    return set_reference_impl(b, rawRef, hrRow0, hrRow1, stream);
}

// B: coarse -> fine tile tracking of the moved pyramid against the reference
static int track_tiles(mfsr_burst* b, const mfsr_prealign* hostBase, mfsr_stream_t stream)
{
    const mfsr_config& c = b->cfg;
    Layout& L = b->L;
    const mfsr_float2 zero2 = {0.0f, 0.0f};
    for (int l = 0; l < c.levels; l++) {
        const int pi = ilog2(c.levelFactor[l]);
        const Img& ref = L.refPyr[pi];
        const Img& mov = L.movPyr[pi];
        const int T = c.tileSize[l], S = c.maxShift[l];
        const int tiles = L.tcx[l] * L.tcy[l];
        const mfsr_float2* pre = nullptr;
        if (c.fused && l > 0) {
            // B8 folded into the tracker: the pre-shifts come straight from the previous level's shifts
            TRY(mfsr_trackTilesFusedUp((const float*)ref.ptr, (const float*)mov.ptr, (const mfsr_float2*)L.shifts[l - 1].ptr,
                                       L.shifts[l - 1].pitch, c.levelFactor[l - 1], c.levelFactor[l], L.tcx[l - 1], L.tcy[l - 1],
                                       c.tileSize[l - 1], (mfsr_float2*)L.shifts[l].ptr, L.shifts[l].pitch, ref.w, ref.h, ref.pitch, S,
                                       T, L.tcx[l], L.tcy[l], c.minimumThreshold, L.refSq[l], c.preAlign ? L.preResult : nullptr,
                                       1.0f / (float)c.levelFactor[l], stream));
            continue;
        }
        if (l > 0) {
            TRY(mfsr_UpSampleShifts((const mfsr_float2*)L.shifts[l - 1].ptr, (mfsr_float2*)L.pre[l].ptr,
                                    L.shifts[l - 1].pitch, L.pre[l].pitch, c.levelFactor[l - 1], c.levelFactor[l],
                                    L.tcx[l - 1], L.tcy[l - 1], L.tcx[l], L.tcy[l], c.tileSize[l - 1], T, stream));
            pre = (const mfsr_float2*)L.pre[l].ptr;
        }
        if (c.fused) {
            TRY(mfsr_trackTilesFusedBase((const float*)ref.ptr, (const float*)mov.ptr, pre, L.pre[l].pitch,
                                         (mfsr_float2*)L.shifts[l].ptr, L.shifts[l].pitch, ref.w, ref.h, ref.pitch, S, T,
                                         L.tcx[l], L.tcy[l], c.minimumThreshold, L.refSq[l],
                                         c.preAlign ? L.preResult : nullptr, 1.0f / (float)c.levelFactor[l], stream));
        } else {
            if (!pre) {
                MFSR_HIP_TRY(hipMemsetAsync(L.pre[l].ptr, 0, (size_t)L.pre[l].pitch * L.pre[l].h, mfsr_s(stream)));
                pre = (const mfsr_float2*)L.pre[l].ptr;
            }
            TRY(mfsr_convertToTilesOverlapBorder((const float*)ref.ptr, L.refTiles, ref.w, ref.h, ref.pitch, S, T, L.tcx[l],
                                                 L.tcy[l], zero2, 0.0f, stream));
            mfsr_float2 base = zero2;
            float rot = 0.0f;
            if (hostBase) {
                const float inv = 1.0f / (float)c.levelFactor[l];
                base.x = hostBase->shiftX * inv;
                base.y = hostBase->shiftY * inv;
                rot = hostBase->rotation;
            }
            TRY(mfsr_convertToTilesOverlapPreShift((const float*)mov.ptr, L.movTiles, pre, L.pre[l].pitch, mov.w, mov.h,
                                                   mov.pitch, S, T, L.tcx[l], L.tcy[l], base, rot, stream));
            TRY(mfsr_crossCorrelateTiles(L.refTiles, L.movTiles, L.cc, S, T, tiles, stream));
            TRY(mfsr_squaredSum(L.refTiles, L.sqsum, S, T, tiles, stream));
            TRY(mfsr_boxFilterWithBorderX(L.movTiles, L.boxX, S, T, tiles, stream));
            TRY(mfsr_boxFilterWithBorderY(L.boxX, L.boxY, S, T, tiles, stream));
            TRY(mfsr_normalizedCC(L.cc, L.sqsum, L.boxY, L.dist, S, T, tiles, stream));
            TRY(mfsr_findMinimum(L.dist, (mfsr_float2*)L.shifts[l].ptr, L.shifts[l].pitch, S, tiles, L.tcx[l],
                                 c.minimumThreshold, stream));
            TRY(mfsr_addRoundedPreShift(pre, L.pre[l].pitch, (mfsr_float2*)L.shifts[l].ptr, L.shifts[l].pitch, L.tcx[l],
                                        L.tcy[l], stream));
        }
    }
    return MFSR_OK;
}

// index of the upload slot (ring slot, or ring + i for reference slot i) that raw points to; -1 for caller-owned frames
static int upload_slot_of(const mfsr_burst* b, const uint16_t* raw)
{
    if (!raw) return -1;
    for (int i = 0; i < b->cfg.uploadRing; i++)
        if (b->L.rawRing[i] == raw) return i;
    for (int i = 0; i < 2; i++)
        if (b->cfg.uploadRing > 0 && b->L.refRaw[i] == raw) return b->cfg.uploadRing + i;
    return -1;
}

static int align_deferred(mfsr_burst* b, mfsr_stream_t stream);

// G: the waiting group (b->pend, 1 .. MFSR_MAX_FUSE_GROUP aligned frames) onto its accumulators (timed with HIP events on request)
static int accumulate_pending(mfsr_burst* b, mfsr_stream_t callerStream);

// a group that was held back for mfsr_burst_finish_host and is fused the ordinary way after all (flush / finish / another
// accumulator pair / its upload slot is needed): as its own launch, exactly as if it had not been held
static int fuse_held(mfsr_burst* b, mfsr_stream_t callerStream)
{
    if (!b->heldHas) return MFSR_OK;
    const mfsr_burst::Pending later = b->pend;
    b->pend = b->held;
    b->heldHas = false;
    b->held.n = 0;
    const int rc = accumulate_pending(b, callerStream);
    b->pend = later;
    return rc;
}

static int accumulate_pending(mfsr_burst* b, mfsr_stream_t callerStream)
{
    TRY(fuse_held(b, callerStream));       // it comes before the frames that arrived after it
    TRY(align_deferred(b, callerStream));  // frames of the group that were waiting for their batch
    TRY(wait_uploads(b, callerStream));    // (a reference frame of the group is read by the fuse only)
    const mfsr_config& c = b->cfg;
    Layout& L = b->L;
    const mfsr_burst::Pending p = b->pend;
    b->pend.n = 0;
    const int n = p.n;
    if (n == 0) return MFSR_OK;
    mfsr_float3* imgOut = p.imgOut;
    mfsr_float3* totalWeights = p.totalWeights;
    // with asyncFuse the launch goes to the burst's own stream, ordered after the alignment of its frames
    mfsr_stream_t stream = callerStream;
    if (b->fuseStream) {
        stream = (mfsr_stream_t)b->fuseStream;
        for (int i = 0; i < n; i++) MFSR_HIP_TRY(hipStreamWaitEvent(b->fuseStream, b->evAligned[p.slot[i]], 0));
    }
    const mfsr_float3 white = {c.white[0], c.white[1], c.white[2]};
    const mfsr_float3 black = {c.black[0], c.black[1], c.black[2]};
    const int strideOut = 12 * L.hrW;
    TRY(mfsr_set_cfa_pattern(c.cfa));
    const bool timed = b->timing && b->nEvents < kMaxTimedLaunches;
    if (timed) {
        const int i = b->nEvents;
        if (!b->evStart[i]) MFSR_HIP_TRY(hipEventCreate(&b->evStart[i]));
        if (!b->evStop[i]) MFSR_HIP_TRY(hipEventCreate(&b->evStop[i]));
        MFSR_HIP_TRY(hipEventRecord(b->evStart[i], mfsr_s(stream)));
    }
    {
        const mfsr_float4* masks[MFSR_MAX_FUSE_GROUP];
        mfsr_tex2d flows[MFSR_MAX_FUSE_GROUP];
        for (int i = 0; i < n; i++) {
            masks[i] = (const mfsr_float4*)p.mask[i]->ptr;
            flows[i] = as_tex(*p.flow[i]);
        }
        // the first fuse after mfsr_burst_begin overwrites the accumulators (they are not zeroed or read)
        const int freshNow = b->fresh.has && b->fresh.imgOut == imgOut && b->fresh.totalWeights == totalWeights;
        if (b->fresh.has && !freshNow) {
            // begin(A, W) followed by a fuse into other accumulators: A and W still owe their zeroes
            const size_t bytes = (size_t)12 * L.hrW * L.hrH;
            MFSR_HIP_TRY(hipMemsetAsync(b->fresh.imgOut, 0, bytes, mfsr_s(stream)));
            MFSR_HIP_TRY(hipMemsetAsync(b->fresh.totalWeights, 0, bytes, mfsr_s(stream)));
        }
        b->fresh.has = false;
        TRY(mfsr_accumulateSuperResFullN(n, p.raw, imgOut, totalWeights, masks, as_tex(L.kparam4), flows, white, black, L.W, L.H,
                                         c.scale, strideOut, p.mask[0]->pitch, freshNow, stream));
    }
    if (timed) {
        MFSR_HIP_TRY(hipEventRecord(b->evStop[b->nEvents], mfsr_s(stream)));
        b->nEvents++;
        b->nFramesTimed += n;
    }
    if (b->copyStream) {
        // upload slots whose raw frame this launch was the last to read may be overwritten once it has run
        for (int j = 0; j < n; j++) {
            const int us = upload_slot_of(b, p.raw[j]);
            if (us >= 0) {
                MFSR_HIP_TRY(hipEventRecord(b->evFree[us], mfsr_s(stream)));
                b->freeRecorded[us] = true;
            }
        }
    }
    if (b->fuseStream) {
        for (int i = 0; i < n; i++) {
            MFSR_HIP_TRY(hipEventRecord(b->evFused[p.slot[i]], b->fuseStream));
            b->fusedOutstanding[p.slot[i]] = true;
        }
    }
    return MFSR_OK;
}

// `stream` is about to read the kernel parameters / the fallback image: wait for the launches that make them
static int wait_ref_products(mfsr_burst* b, mfsr_stream_t stream)
{
    if (b->refOnFuse && mfsr_s(stream) != b->fuseStream) MFSR_HIP_TRY(hipStreamWaitEvent(mfsr_s(stream), b->evRefDone, 0));
    return MFSR_OK;
}

// the caller's stream waits for every fuse issued so far, and for the reference products (no-op without asyncFuse)
static int join_fuse(mfsr_burst* b, mfsr_stream_t stream)
{
    if (!b->fuseStream) return MFSR_OK;
    TRY(wait_ref_products(b, stream));
    for (int i = 0; i < kRing; i++)
        if (b->fusedOutstanding[i]) {
            MFSR_HIP_TRY(hipStreamWaitEvent(mfsr_s(stream), b->evFused[i], 0));
            b->fusedOutstanding[i] = false;
        }
    return MFSR_OK;
}

// fuse the frames still waiting for the rest of their group, then join: afterwards the caller's stream sees every frame
static int flush_pending(mfsr_burst* b, mfsr_stream_t stream, bool materializeFresh)
{
    if (materializeFresh && b->fresh.has && b->pend.n == 0 && !b->heldHas) {
        // mfsr_burst_begin with no frame fused since: the accumulators must read as zero
        const size_t bytes = (size_t)12 * b->L.hrW * b->L.hrH;
        MFSR_HIP_TRY(hipMemsetAsync(b->fresh.imgOut, 0, bytes, mfsr_s(stream)));
        MFSR_HIP_TRY(hipMemsetAsync(b->fresh.totalWeights, 0, bytes, mfsr_s(stream)));
        b->fresh.has = false;
    }
    TRY(accumulate_pending(b, stream));
    return join_fuse(b, stream);
}

// A1 + (I) + B + D + F of one frame into ring slot `slot`: *flowOut / *maskOut name the buffers that hold the result
// phases: ALIGN_PRE = everything up to the flow field (+ first warp), ALIGN_LK = the Lucas-Kanade iterations, ALIGN_POST = the
// robustness mask.  align_group runs PRE and POST frame by frame and the iterations of a whole group in one launch each.
enum { ALIGN_PRE = 1, ALIGN_LK = 2, ALIGN_POST = 4, ALIGN_ALL = 7 };
static bool lk_warped_path(const mfsr_burst* b)
{
    // fused LK: the warped moved image travels from launch to launch (every pixel warped once per iteration, by the
    // thread that has just updated its flow) instead of being re-gathered for every tile halo; MFSR_LK_WARPED=0: A/B
    static const bool lkWarped = [] {
        const char* e = getenv("MFSR_LK_WARPED");
        return !(e && e[0] == '0');
    }();
    const mfsr_config& c = b->cfg;
    return c.fused && lkWarped && c.lkIterations > 0 && b->L.tw >= 64 && b->L.th >= 32;
}

static int align_frame(mfsr_burst* b, const uint16_t* raw, int isReference, int slot, Img** flowOut, Img** maskOut,
                       mfsr_stream_t stream, int phases = ALIGN_ALL)
{
    const mfsr_config& c = b->cfg;
    Layout& L = b->L;
    TRY(wait_uploads(b, stream));
    // the slot's buffers are free once the fuse that read them last has run
    if ((phases & ALIGN_PRE) && b->fuseStream && b->fusedOutstanding[slot]) {
        MFSR_HIP_TRY(hipStreamWaitEvent(mfsr_s(stream), b->evFused[slot], 0));
        b->fusedOutstanding[slot] = false;
    }
    Img* flow = &L.flowBuf[2 * slot];
    Img* other = &L.flowBuf[2 * slot + 1];
    Img* mask = &L.maskBuf[slot];
    const bool warped = lk_warped_path(b);
    if (isReference) {
        // identity flow, certainty 1
        if (phases & ALIGN_PRE) {
            MFSR_HIP_TRY(hipMemsetAsync(flow->ptr, 0, (size_t)flow->pitch * flow->h, mfsr_s(stream)));
            TRY(mfsr_fill_f32((float*)mask->ptr, (size_t)mask->pitch / 4 * mask->h, 1.0f, stream));
        }
    } else {
      if (phases & ALIGN_PRE) {
        if (!b->movPrepared) TRY(prepare_frame(b, raw, L.movHalf, L.movPyr, stream));
        // I: global pre-alignment (base shift + rotation of this frame against the reference), kept in device memory
        mfsr_prealign hostBase;
        const mfsr_prealign* hb = nullptr;
        if (c.preAlign) {
            if (!b->movPrepared)
                TRY(mfsr_preAlignPyramid((const float*)L.movPyr[0].ptr, L.tw, L.th, L.movPyr[0].pitch, L.preMovPyr, stream));
            TRY(mfsr_preAlign(L.preRefPyr, L.preMovPyr, L.tw, L.th, c.preAlignMaxAngle, L.preWork, L.preResult, stream));
            b->preCur = L.preResult;  // (inside align_deferred this names the frame's align set: swap_set)
            if (!c.fused) {
                // the reference-shaped entry points take baseShift / baseRotation by value: one host round trip
                MFSR_HIP_TRY(hipMemcpyAsync(&hostBase, L.preResult, sizeof(hostBase), hipMemcpyDeviceToHost, mfsr_s(stream)));
                MFSR_HIP_TRY(hipStreamSynchronize(mfsr_s(stream)));
                hb = &hostBase;
            }
        }
        if (!b->givenShifts) TRY(track_tiles(b, hb, stream));
        const int last = c.levels - 1;
        const Img& tileShifts = b->givenShifts ? *b->givenShifts : L.shifts[last];
        const mfsr_float2 zero2 = {0.0f, 0.0f};
        if (warped) {
            TRY(mfsr_CreateFlowFieldWarped((mfsr_float2*)flow->ptr, as_tex(tileShifts), L.tw, L.th, flow->pitch, zero2, 0.0f,
                                           c.preAlign ? L.preResult : nullptr, (const float*)L.refPyr[0].ptr,
                                           (const float*)L.movPyr[0].ptr, L.refPyr[0].pitch, (float*)L.lkSum[0].ptr,
                                           (float*)L.lkDiff[0].ptr, L.lkSum[0].pitch, stream));
        } else if (c.preAlign && c.fused) {
            TRY(mfsr_CreateFlowFieldFromTilesBase((mfsr_float2*)flow->ptr, as_tex(tileShifts), L.tw, L.th, flow->pitch,
                                                  L.preResult, stream));
        } else {
            mfsr_float2 base = zero2;
            float rot = 0.0f;
            if (hb) {
                base.x = hb->shiftX;
                base.y = hb->shiftY;
                rot = hb->rotation;
            }
            TRY(mfsr_CreateFlowFieldFromTiles((mfsr_float2*)flow->ptr, as_tex(tileShifts), c.tileSize[last], L.tcx[last],
                                              L.tcy[last], L.tw, L.th, flow->pitch, base, rot, stream));
        }
      }
      if (phases & ALIGN_LK) {
        for (int it = 0; it < c.lkIterations; it++) {
            if (warped) {
                const bool lastIt = it == c.lkIterations - 1;
                const int in = it & 1, out = in ^ 1;
                // the sweep kernel wherever it applies, also for a single frame: the frame-by-frame paths (frame streams,
                // joint mode, mfsr_burst_align_frame of the multi-GPU layer, groups of one) then give the very bits of the
                // frame-batched burst -- its result does not depend on how many frames share a launch
                mfsr_lk_frame one;
                one.shiftsIn = (const mfsr_float2*)flow->ptr;
                one.shiftsOut = (mfsr_float2*)other->ptr;
                one.movedImg = (const float*)L.movPyr[0].ptr;
                one.sumIn = (const float*)L.lkSum[in].ptr;
                one.diffIn = (const float*)L.lkDiff[in].ptr;
                one.sumOut = lastIt ? nullptr : (float*)L.lkSum[out].ptr;
                one.diffOut = lastIt ? nullptr : (float*)L.lkDiff[out].ptr;
                const int rcs = mfsr_lucasKanadeSweepBatch(1, &one, (const float*)L.refPyr[0].ptr, flow->pitch, L.refPyr[0].pitch,
                                                           L.lkSum[0].pitch, L.tw, L.th, c.lkHalfWindow, c.lkMinDet,
                                                           lastIt ? (float)L.flowScale : 1.0f, stream);
                if (rcs != MFSR_E_UNSUPPORTED) {
                    if (rcs) return rcs;
                    Img* t = flow;
                    flow = other;
                    other = t;
                    continue;
                }
                TRY(mfsr_lucasKanadeIterationWarped((const mfsr_float2*)flow->ptr, (mfsr_float2*)other->ptr, flow->pitch,
                                                    (const float*)L.refPyr[0].ptr, (const float*)L.movPyr[0].ptr, L.refPyr[0].pitch,
                                                    (const float*)L.lkSum[in].ptr, (const float*)L.lkDiff[in].ptr,
                                                    lastIt ? nullptr : (float*)L.lkSum[out].ptr,
                                                    lastIt ? nullptr : (float*)L.lkDiff[out].ptr, L.lkSum[0].pitch, L.tw, L.th,
                                                    c.lkHalfWindow, c.lkMinDet, lastIt ? (float)L.flowScale : 1.0f, stream));
                Img* t = flow;
                flow = other;
                other = t;
            } else if (c.fused) {
                TRY(mfsr_lucasKanadeIterationFused((const mfsr_float2*)flow->ptr, (mfsr_float2*)other->ptr, flow->pitch,
                                                   (const float*)L.refPyr[0].ptr, (const float*)L.movPyr[0].ptr,
                                                   L.refPyr[0].pitch, L.tw, L.th, c.lkHalfWindow, c.lkMinDet,
                                                   it == c.lkIterations - 1 ? (float)L.flowScale : 1.0f, stream));
                Img* t = flow;
                flow = other;
                other = t;
            } else {
                TRY(mfsr_WarpingKernel(L.tw, L.th, L.warped.pitch, as_tex(*flow), (float*)L.warped.ptr, as_tex(L.movPyr[0]),
                                       stream));
                // texSource = warped moved frame, texTarget = reference: the reference's stencil is
                // minus the usual derivative (opticalFlow.cu:116-119) and Iz = source - target (:131),
                // so this is the order for which `shift += UV` (:322-323) descends.
                TRY(mfsr_ComputeDerivativesKernel(L.tw, L.th, L.Ix.pitch, (float*)L.Ix.ptr, (float*)L.Iy.ptr,
                                                  (float*)L.It.ptr, as_tex(L.warped), as_tex(L.refPyr[0]), stream));
                TRY(mfsr_lucasKanadeOptim((mfsr_float2*)flow->ptr, (const float*)L.Ix.ptr, (const float*)L.Iy.ptr,
                                          (const float*)L.It.ptr, flow->pitch, L.Ix.pitch, L.tw, L.th, c.lkHalfWindow,
                                          c.lkMinDet, stream));
            }
        }
        if (L.flowScale != 1 && !(c.fused && c.lkIterations > 0))  // the fused LK scales on its last iteration
            TRY(mfsr_scaleFlow((mfsr_float2*)flow->ptr, flow->pitch, L.tw, L.th, (float)L.flowScale, stream));
      } else if (c.fused && (c.lkIterations & 1)) {
        // the iterations ran elsewhere (align_group): an odd number of them leaves the flow in the slot's other buffer
        Img* t = flow;
        flow = other;
        other = t;
      }
      if (phases & ALIGN_POST) {
        // F: robustness mask (the 1-px ring is never written by the kernel -> zero it)
        if (c.fused) {
            TRY(mfsr_robustnessMaskFused((const mfsr_float3*)L.refHalf.ptr, (const mfsr_float3*)L.movHalf.ptr,
                                         (mfsr_float4*)mask->ptr, as_tex(*flow), L.hw, L.hh, L.refHalf.pitch, mask->pitch, c.alpha,
                                         c.beta, c.thresholdM, stream));
        } else {
            TRY(mfsr_zeroRing_f32x4((mfsr_float4*)mask->ptr, mask->pitch, L.hw, L.hh, stream));
            TRY(mfsr_ComputeRobustnessMask((const mfsr_float3*)L.refHalf.ptr, (const mfsr_float3*)L.movHalf.ptr,
                                           (mfsr_float4*)mask->ptr, as_tex(*flow), L.hw, L.hh, L.refHalf.pitch, mask->pitch,
                                           c.alpha, c.beta, c.thresholdM, stream));
        }
      }
    }
    *flowOut = flow;
    *maskOut = mask;
    b->slotFlow[slot] = flow;  // mfsr_burst_debug_frame_views
    return MFSR_OK;
}

// Frame-batched alignment.  The frames of a fuse group are independent given the reference's products, and one frame's
// alignment is a chain of small launches (prepare, two tracker levels, flow field, lkIterations x Lucas-Kanade, robustness:
// 8 at the defaults, 10..30 us each at 4K) -- so add_frame only REGISTERS a frame while its group fills, and the group is
// aligned as one batch when it is complete (or on flush / finish): the per-frame stages frame after frame into per-frame
// intermediates (Layout::sets), every Lucas-Kanade iteration of all frames in ONE launch (mfsr_lucasKanadeSweepBatch).
// Same kernels' arithmetic per frame as the frame-by-frame path except the sweep kernel's row-sum order (fp32 rounding of
// the flow).  Frames whose intermediates the caller supplies (frame streams, the joint mode) take the frame-by-frame path.
static bool can_defer_alignment(const mfsr_burst* b)
{
    static const bool on = [] {
        const char* e = getenv("MFSR_ALIGN_BATCH");
        return !(e && e[0] == '0');
    }();
    const mfsr_config& c = b->cfg;
    // (host bursts: frames are aligned one by one as they come off the PCIe link -- waiting for a whole group would leave
    // the GPU idle during the first uploads and would put the whole last group's alignment between the last upload and
    // the first byte of the download; MFSR_HOST_DEFER=1 batches the groups after the first anyway)
    static const bool hostDefer = [] {
        const char* e = getenv("MFSR_HOST_DEFER");
        return e && e[0] == '1';
    }();
    // ... and a host burst that was enqueued while the previous one was still being fused and downloaded (bursts back to
    // back: b->hostBusy) batches every group: the GPU has work queued, so frame-by-frame launches buy no latency and cost
    // a tenth of the alignment's throughput
    const bool hostImmediate = b->holdLastGroup && !b->hostBusy && (b->framesSinceRef < b->group || !hostDefer);
    return on && b->group > 1 && b->L.nSets >= b->group - 1 && lk_warped_path(b) && c.lkHalfWindow >= 1 && c.lkHalfWindow <= 7 &&
           !b->movPrepared && !b->givenShifts && !hostImmediate;
}

static int align_deferred(mfsr_burst* b, mfsr_stream_t stream)
{
    const mfsr_config& c = b->cfg;
    Layout& L = b->L;
    int idx[MFSR_MAX_FUSE_GROUP], n = 0;
    for (int i = 0; i < b->pend.n; i++)
        if (b->pend.deferred[i]) idx[n++] = i;
    if (n == 0) return MFSR_OK;
    TRY(wait_uploads(b, stream));
    // the moved-frame intermediates of batch position q (0: the Layout's own members, q > 0: align set q - 1)
    auto movHalf = [&](int q) -> Img& { return q == 0 ? L.movHalf : L.sets[q - 1].movHalf; };
    auto movPyr = [&](int q) -> Img* { return q == 0 ? L.movPyr : L.sets[q - 1].movPyr; };
    auto shiftsOf = [&](int q) -> Img* { return q == 0 ? L.shifts : L.sets[q - 1].shifts; };
    auto lkSumOf = [&](int q) -> Img* { return q == 0 ? L.lkSum : L.sets[q - 1].lkSum; };
    auto lkDiffOf = [&](int q) -> Img* { return q == 0 ? L.lkDiff : L.sets[q - 1].lkDiff; };
    // Every per-frame stage as ONE launch over the batch (gridDim.z = frame) where the default kernels apply: the fused prepare
    // kernel, the compile-time tracker at every level, the flow field + first warp, later the fused robustness kernel.
    // Anything else (monochrome frames, pre-alignment, other tile sizes) runs the same stages frame by frame.
    static const bool stageBatchOn = [] {
        const char* e = getenv("MFSR_ALIGN_BATCH");
        return !(e && e[0] == '1' && e[1] == 's');  // MFSR_ALIGN_BATCH=1s: batch the Lucas-Kanade launches only (A/B)
    }();
    const int nlev = ilog2(L.maxFactor) + 1;
    bool stageBatch = stageBatchOn && c.fused && !c.mono && !c.preAlign && b->ntaps / 2 <= 8 && L.tw == L.hw && L.th == L.hh;
    for (int l = 0; l < c.levels && stageBatch; l++) stageBatch = mfsr_trackTilesFastSupported(c.tileSize[l], c.maxShift[l]) != 0;
    int mq[MFSR_MAX_FUSE_GROUP], m = 0;  // batch positions of the moved (non-reference) frames
    for (int q = 0; q < n; q++)
        if (!b->pend.isRef[idx[q]]) mq[m++] = q;
    if (stageBatch) {
        for (int q = 0; q < n; q++) {
            const int i = idx[q], slot = b->pend.slot[i];
            if (b->pend.isRef[i]) {  // identity flow, certainty 1 (and the slot wait) through the frame path
                Img *flow = nullptr, *mask = nullptr;
                TRY(align_frame(b, b->pend.raw[i], 1, slot, &flow, &mask, stream, ALIGN_PRE));
            } else if (b->fuseStream && b->fusedOutstanding[slot]) {
                // the slot's buffers are free once the fuse that read them last has run
                MFSR_HIP_TRY(hipStreamWaitEvent(mfsr_s(stream), b->evFused[slot], 0));
                b->fusedOutstanding[slot] = false;
            }
        }
        if (m > 0) {
            TRY(mfsr_set_cfa_pattern(c.cfa));
            // A1 + luma + prefilter + first pyramid level
            mfsr_prepare_frame pf[MFSR_MAX_FUSE_GROUP];
            for (int k = 0; k < m; k++) {
                Img* pyr = movPyr(mq[k]);
                pf[k].dataIn = b->pend.raw[idx[mq[k]]];
                pf[k].halfOut = (mfsr_float3*)movHalf(mq[k]).ptr;
                pf[k].pyr0 = (float*)pyr[0].ptr;
                pf[k].pyr1 = nlev > 1 ? (float*)pyr[1].ptr : nullptr;
            }
            TRY(mfsr_prepareFrameFusedBatch(m, pf, L.movHalf.pitch, c.maxVal, L.hw, L.hh, L.movPyr[0].pitch, nlev > 1 ? L.movPyr[1].pitch : 0,
                                            b->taps, b->ntaps, stream));
            for (int k = 0; k < m; k++) {
                Img* pyr = movPyr(mq[k]);
                for (int i = 2; i < nlev; i++)
                    TRY(mfsr_downsample2x((const float*)pyr[i - 1].ptr, pyr[i - 1].pitch, (float*)pyr[i].ptr, pyr[i].pitch, pyr[i].w, pyr[i].h,
                                          stream));
            }
            // B: coarse -> fine, one launch per level
            for (int l = 0; l < c.levels; l++) {
                const int pi = ilog2(c.levelFactor[l]);
                mfsr_track_frame tf[MFSR_MAX_FUSE_GROUP];
                for (int k = 0; k < m; k++) {
                    tf[k].movedImg = (const float*)movPyr(mq[k])[pi].ptr;
                    tf[k].coarseShifts = l > 0 ? (const mfsr_float2*)shiftsOf(mq[k])[l - 1].ptr : nullptr;
                    tf[k].coordinates = (mfsr_float2*)shiftsOf(mq[k])[l].ptr;
                    tf[k].base = nullptr;
                }
                const Img& ref = L.refPyr[pi];
                TRY(mfsr_trackTilesFusedBatch(m, tf, (const float*)ref.ptr, l > 0 ? L.shifts[l - 1].pitch : 0, l > 0 ? c.levelFactor[l - 1] : 0,
                                              c.levelFactor[l], l > 0 ? L.tcx[l - 1] : 0, l > 0 ? L.tcy[l - 1] : 0,
                                              l > 0 ? c.tileSize[l - 1] : 0, L.shifts[l].pitch, ref.w, ref.h, ref.pitch, c.maxShift[l],
                                              c.tileSize[l], L.tcx[l], L.tcy[l], c.minimumThreshold, L.refSq[l],
                                              1.0f / (float)c.levelFactor[l], stream));
            }
            // D1 + the first warp
            const int last = c.levels - 1;
            mfsr_flowfield_frame ff[MFSR_MAX_FUSE_GROUP];
            for (int k = 0; k < m; k++) {
                const int slot = b->pend.slot[idx[mq[k]]];
                ff[k].outImg = (mfsr_float2*)L.flowBuf[2 * slot].ptr;
                ff[k].tileShifts = (const mfsr_float2*)shiftsOf(mq[k])[last].ptr;
                ff[k].base = nullptr;
                ff[k].movedImg = (const float*)movPyr(mq[k])[0].ptr;
                ff[k].sumOut = (float*)lkSumOf(mq[k])[0].ptr;
                ff[k].diffOut = (float*)lkDiffOf(mq[k])[0].ptr;
            }
            TRY(mfsr_CreateFlowFieldWarpedBatch(m, ff, L.shifts[last].pitch, L.shifts[last].w, L.shifts[last].h, L.tw, L.th,
                                                L.flowBuf[0].pitch, (const float*)L.refPyr[0].ptr, L.refPyr[0].pitch, L.lkSum[0].pitch, stream));
        }
    } else {
    // per-frame stages up to the flow field + first warp; frame q > 0 of the batch works in align set q - 1
    for (int q = 0; q < n; q++) {
        const int i = idx[q];
        if (q > 0) swap_set(L, L.sets[q - 1]);
        Img *flow = nullptr, *mask = nullptr;
        const int rc = align_frame(b, b->pend.raw[i], b->pend.isRef[i], b->pend.slot[i], &flow, &mask, stream, ALIGN_PRE);
        if (q > 0) swap_set(L, L.sets[q - 1]);
        if (rc) return rc;
    }
    }
    // every Lucas-Kanade iteration of the batch in one launch
    bool batched = true;
    for (int it = 0; it < c.lkIterations && batched; it++) {
        mfsr_lk_frame fr[MFSR_MAX_FUSE_GROUP];
        int m = 0;
        const bool lastIt = it == c.lkIterations - 1;
        const int in = it & 1, out = in ^ 1;
        for (int q = 0; q < n; q++) {
            const int i = idx[q];
            if (b->pend.isRef[i]) continue;
            const int slot = b->pend.slot[i];
            const Img* sum = q == 0 ? L.lkSum : L.sets[q - 1].lkSum;
            const Img* diff = q == 0 ? L.lkDiff : L.sets[q - 1].lkDiff;
            const Img& mov = q == 0 ? L.movPyr[0] : L.sets[q - 1].movPyr[0];
            fr[m].shiftsIn = (const mfsr_float2*)L.flowBuf[2 * slot + in].ptr;
            fr[m].shiftsOut = (mfsr_float2*)L.flowBuf[2 * slot + out].ptr;
            fr[m].movedImg = (const float*)mov.ptr;
            fr[m].sumIn = (const float*)sum[in].ptr;
            fr[m].diffIn = (const float*)diff[in].ptr;
            fr[m].sumOut = lastIt ? nullptr : (float*)sum[out].ptr;
            fr[m].diffOut = lastIt ? nullptr : (float*)diff[out].ptr;
            m++;
        }
        if (m == 0) break;
        const int rc = mfsr_lucasKanadeSweepBatch(m, fr, (const float*)L.refPyr[0].ptr, L.flowBuf[0].pitch, L.refPyr[0].pitch,
                                                  L.lkSum[0].pitch, L.tw, L.th, c.lkHalfWindow, c.lkMinDet,
                                                  lastIt ? (float)L.flowScale : 1.0f, stream);
        if (rc == MFSR_E_UNSUPPORTED && it == 0)
            batched = false;  // (cannot happen after can_defer_alignment; kept as a safe fallback)
        else if (rc)
            return rc;
    }
    // robustness masks (and, without the batch kernel, the iterations frame by frame)
    bool maskBatch = stageBatch && batched && m > 0;
    if (maskBatch) {
        mfsr_robustness_frame rf[MFSR_MAX_FUSE_GROUP];
        for (int k = 0; k < m; k++) {
            const int slot = b->pend.slot[idx[mq[k]]];
            rf[k].movedHalf = (const mfsr_float3*)movHalf(mq[k]).ptr;
            rf[k].mask = (mfsr_float4*)L.maskBuf[slot].ptr;
            rf[k].flow = (const mfsr_float2*)L.flowBuf[2 * slot + (c.lkIterations & 1)].ptr;  // where the ping-pong ends
        }
        const int rc = mfsr_robustnessMaskFusedBatch(m, rf, (const mfsr_float3*)L.refHalf.ptr, L.flowBuf[0].pitch, L.tw, L.th, L.hw, L.hh,
                                                     L.refHalf.pitch, L.maskBuf[0].pitch, c.alpha, c.beta, c.thresholdM, stream);
        if (rc == MFSR_E_UNSUPPORTED)
            maskBatch = false;  // (the straight robustness kernel was selected: frame by frame below)
        else if (rc)
            return rc;
    }
    for (int q = 0; q < n; q++) {
        const int i = idx[q];
        if (q > 0) swap_set(L, L.sets[q - 1]);
        Img *flow = nullptr, *mask = nullptr;
        const int rc = align_frame(b, b->pend.raw[i], b->pend.isRef[i], b->pend.slot[i], &flow, &mask, stream,
                                   maskBatch ? 0 : (batched ? ALIGN_POST : (ALIGN_LK | ALIGN_POST)));
        if (q > 0) swap_set(L, L.sets[q - 1]);
        if (rc) return rc;
        b->pend.flow[i] = flow;
        b->pend.mask[i] = mask;
        b->pend.deferred[i] = false;
        b->flowCur = flow;
        b->maskCur = mask;
        if (b->fuseStream) MFSR_HIP_TRY(hipEventRecord(b->evAligned[b->pend.slot[i]], mfsr_s(stream)));
    }
    return MFSR_OK;
}

// hostBurst: the frame belongs to a host-frame burst (mfsr_burst_add_frame_host), whose last groups wait for
// mfsr_burst_finish_host; a frame added through mfsr_burst_add_frame never leaves a complete group waiting
static int add_frame_impl(mfsr_burst* b, const uint16_t* raw, int isReference, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                          bool hostBurst, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && raw && imgOut && totalWeights);
    MFSR_REQUIRE(b->haveRef);
    MFSR_REQUIRE(!b->refStale);  // after mfsr_burst_process_joint: call mfsr_burst_set_reference first
    const mfsr_config& c = b->cfg;
    TRY(mfsr_set_cfa_pattern(c.cfa));
    b->holdLastGroup = hostBurst;
    // G: accumulate onto the HR grid -- alone, or together with the frames that were waiting for the rest of their group
    if (b->pend.n && (b->pend.imgOut != imgOut || b->pend.totalWeights != totalWeights)) TRY(flush_pending(b, stream));
    // a complete group was held back for mfsr_burst_finish_host and frames keep arriving (more than cfg.frames per
    // reference, or a resident frame after a host burst): it is fused the ordinary way before this frame is registered
    if (b->pend.n >= b->group) TRY(accumulate_pending(b, stream));
    MFSR_REQUIRE(b->pend.n < MFSR_MAX_FUSE_GROUP);
    const int slot = b->frameCounter++ % kRing;
    Img *flow = nullptr, *mask = nullptr;
    const bool defer = can_defer_alignment(b);
    if (!defer) {
        TRY(align_deferred(b, stream));  // keep the alignment in frame order on the stream
        TRY(align_frame(b, raw, isReference, slot, &flow, &mask, stream));
        b->flowCur = flow;
        b->maskCur = mask;
        if (b->fuseStream) MFSR_HIP_TRY(hipEventRecord(b->evAligned[slot], mfsr_s(stream)));
    }
    if (defer) b->slotFlow[slot] = nullptr;  // (mfsr_burst_debug_frame_views: not aligned yet)
    const int i = b->pend.n++;
    b->pend.slot[i] = slot;
    b->pend.raw[i] = raw;
    b->pend.flow[i] = flow;
    b->pend.mask[i] = mask;
    b->pend.deferred[i] = defer;
    b->pend.isRef[i] = isReference;
    b->pend.imgOut = imgOut;
    b->pend.totalWeights = totalWeights;
    b->framesSinceRef++;
    if (b->pend.n < b->group) return MFSR_OK;  // the group is fused when its last frame arrives (or on flush / finish)
    // host bursts: the group that the burst's last frame completes is fused band by band by mfsr_burst_finish_host, so that
    // finished bands of the image leave for the host while the later ones are still being fused
    // (only the group the burst's LAST frame completes: frames beyond cfg.frames are fused as they come)
    if (b->holdLastGroup && b->framesSinceRef == c.frames) return MFSR_OK;
    // ... and so does the group before it (MFSR_HOST_HOLD=2, the default): the first band of the image is then complete after
    // 1/8 of two groups' fuse instead of after a whole group's, and the download -- the tail of the burst -- starts that much
    // earlier; the earlier groups are fused as they arrive, under the uploads
    static const int holdGroups = [] {
        const char* e = getenv("MFSR_HOST_HOLD");
        return e ? atoi(e) : 2;
    }();
    if (b->holdLastGroup && holdGroups >= 2 && !b->heldHas && c.frames - b->framesSinceRef <= b->group &&
        c.frames - b->framesSinceRef > 0 && !b->pend.deferred[0]) {
        bool aligned = true;
        for (int j = 0; j < b->pend.n; j++) aligned = aligned && !b->pend.deferred[j];
        if (aligned) {
            b->held = b->pend;
            b->heldHas = true;
            b->pend.n = 0;
            return MFSR_OK;
        }
    }
    return accumulate_pending(b, stream);
}

extern "C" int mfsr_burst_add_frame(mfsr_burst* b, const uint16_t* raw, int isReference, mfsr_float3* imgOut,
                                    mfsr_float3* totalWeights, mfsr_stream_t stream)
{
    return add_frame_impl(b, raw, isReference, imgOut, totalWeights, false, stream);
}

extern "C" int mfsr_burst_group_size(const mfsr_config* cfg)
{
    if (!cfg || cfg->pairFrames <= 0) return 1;
    if (cfg->pairFrames >= 2) return cfg->pairFrames < MFSR_MAX_FUSE_GROUP ? cfg->pairFrames : MFSR_MAX_FUSE_GROUP;
    // as many as one launch takes: the x2 and x4 Bayer tile kernels fuse four, every other geometry two
    return ((cfg->scale == 2 || cfg->scale == 4) && !cfg->mono) ? MFSR_MAX_FUSE_GROUP : 2;
}

// ---- building blocks of stripe-sharded bursts (multi-GPU: frames are aligned where they live, every rank fuses ALL
//      frames onto its own stripe of HR rows; csrc/dist.cpp and the Python mirror drive these) -----------------------
extern "C" int mfsr_burst_field_dims(const mfsr_burst* b, int* flowW, int* flowH, int* maskW, int* maskH)
{
    MFSR_REQUIRE(b != nullptr);
    if (flowW) *flowW = b->L.tw;
    if (flowH) *flowH = b->L.th;
    if (maskW) *maskW = b->L.hw;
    if (maskH) *maskH = b->L.hh;
    return MFSR_OK;
}

extern "C" int mfsr_burst_align_frame(mfsr_burst* b, const uint16_t* raw, int isReference, mfsr_float2* flowOut, int flowPitch,
                                      mfsr_float4* maskOut, int maskPitch, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && raw && flowOut && maskOut);
    MFSR_REQUIRE(b->haveRef);
    MFSR_REQUIRE(!b->refStale);  // after mfsr_burst_process_joint: call mfsr_burst_set_reference first
    Layout& L = b->L;
    MFSR_REQUIRE((long long)flowPitch >= 8LL * L.tw && (flowPitch & 7) == 0 && ((uintptr_t)flowOut & 7) == 0);
    MFSR_REQUIRE((long long)maskPitch >= 16LL * L.hw && (maskPitch & 15) == 0 && ((uintptr_t)maskOut & 15) == 0);
    TRY(mfsr_set_cfa_pattern(b->cfg.cfa));
    const int slot = b->frameCounter++ % kRing;
    Img *flow = nullptr, *mask = nullptr;
    TRY(align_frame(b, raw, isReference, slot, &flow, &mask, stream));
    b->flowCur = flow;
    b->maskCur = mask;
    MFSR_HIP_TRY(hipMemcpy2DAsync(flowOut, flowPitch, flow->ptr, flow->pitch, (size_t)L.tw * 8, L.th, hipMemcpyDeviceToDevice,
                                  mfsr_s(stream)));
    MFSR_HIP_TRY(hipMemcpy2DAsync(maskOut, maskPitch, mask->ptr, mask->pitch, (size_t)L.hw * 16, L.hh, hipMemcpyDeviceToDevice,
                                  mfsr_s(stream)));
    return MFSR_OK;
}

// mfsr_burst_align_frame for several frames: groups of up to mfsr_burst_group_size frames are aligned as one batch (the
// launches of mfsr_burst_add_frame's groups: a frame's result does not depend on the batch it is in)
extern "C" int mfsr_burst_align_frames(mfsr_burst* b, int nFrames, const uint16_t* const* raws, const int* isReference,
                                       mfsr_float2* const* flowOut, int flowPitch, mfsr_float4* const* maskOut, int maskPitch,
                                       mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && raws && flowOut && maskOut && nFrames >= 0);
    MFSR_REQUIRE(b->haveRef && !b->refStale);
    MFSR_REQUIRE(b->pend.n == 0);  // no frame of an add_frame group may be waiting
    Layout& L = b->L;
    MFSR_REQUIRE((long long)flowPitch >= 8LL * L.tw && (flowPitch & 7) == 0);
    MFSR_REQUIRE((long long)maskPitch >= 16LL * L.hw && (maskPitch & 15) == 0);
    for (int k = 0; k < nFrames; k++)
        MFSR_REQUIRE(raws[k] && flowOut[k] && maskOut[k] && ((uintptr_t)flowOut[k] & 7) == 0 && ((uintptr_t)maskOut[k] & 15) == 0);
    TRY(mfsr_set_cfa_pattern(b->cfg.cfa));
    const bool batch = can_defer_alignment(b);
    const int per = batch ? b->group : 1;
    for (int k0 = 0; k0 < nFrames; k0 += per) {
        const int n = nFrames - k0 < per ? nFrames - k0 : per;
        for (int j = 0; j < n; j++) {
            const int isRef = isReference ? isReference[k0 + j] : 0;
            const int slot = b->frameCounter++ % kRing;
            b->pend.slot[j] = slot;
            b->pend.raw[j] = raws[k0 + j];
            b->pend.isRef[j] = isRef;
            b->pend.deferred[j] = batch;
            b->pend.flow[j] = nullptr;
            b->pend.mask[j] = nullptr;
            if (!batch) {
                const int rc = align_frame(b, raws[k0 + j], isRef, slot, &b->pend.flow[j], &b->pend.mask[j], stream);
                if (rc) {
                    b->pend.n = 0;
                    return rc;
                }
            }
        }
        b->pend.n = n;
        const int rc = align_deferred(b, stream);
        const mfsr_burst::Pending p = b->pend;
        b->pend.n = 0;  // these frames are not waiting for a fuse
        if (rc) return rc;
        for (int j = 0; j < n; j++) {
            b->flowCur = p.flow[j];
            b->maskCur = p.mask[j];
            MFSR_HIP_TRY(hipMemcpy2DAsync(flowOut[k0 + j], flowPitch, p.flow[j]->ptr, p.flow[j]->pitch, (size_t)L.tw * 8, L.th,
                                          hipMemcpyDeviceToDevice, mfsr_s(stream)));
            MFSR_HIP_TRY(hipMemcpy2DAsync(maskOut[k0 + j], maskPitch, p.mask[j]->ptr, p.mask[j]->pitch, (size_t)L.hw * 16, L.hh,
                                          hipMemcpyDeviceToDevice, mfsr_s(stream)));
        }
    }
    return MFSR_OK;
}

extern "C" int mfsr_burst_fuse_rows(mfsr_burst* b, int nFrames, const uint16_t* const* raws, const mfsr_float2* const* flows,
                                    int flowPitch, const mfsr_float4* const* masks, int maskPitch, mfsr_float3* imgOut,
                                    mfsr_float3* totalWeights, int accumulatorsUndefined, int rowBegin, int rowEnd,
                                    mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && raws && flows && masks && imgOut && totalWeights && nFrames >= 1 && nFrames <= MFSR_MAX_FUSE_GROUP);
    MFSR_REQUIRE(b->haveRef);
    TRY(wait_ref_products(b, stream));
    const mfsr_config& c = b->cfg;
    Layout& L = b->L;
    const mfsr_float3 white = {c.white[0], c.white[1], c.white[2]};
    const mfsr_float3 black = {c.black[0], c.black[1], c.black[2]};
    TRY(mfsr_set_cfa_pattern(c.cfa));
    mfsr_tex2d sh[MFSR_MAX_FUSE_GROUP];
    for (int n = 0; n < nFrames; n++) {
        sh[n].ptr = flows[n];
        sh[n].pitch = flowPitch;
        sh[n].width = L.tw;
        sh[n].height = L.th;
    }
    const bool timed = b->timing && b->nEvents < kMaxTimedLaunches;
    if (timed) {
        const int i = b->nEvents;
        if (!b->evStart[i]) MFSR_HIP_TRY(hipEventCreate(&b->evStart[i]));
        if (!b->evStop[i]) MFSR_HIP_TRY(hipEventCreate(&b->evStop[i]));
        MFSR_HIP_TRY(hipEventRecord(b->evStart[i], mfsr_s(stream)));
    }
    TRY(mfsr_accumulateSuperResFullRows(nFrames, raws, imgOut, totalWeights, masks, as_tex(L.kparam4), sh, white, black, L.W, L.H,
                                        c.scale, 12 * L.hrW, maskPitch, accumulatorsUndefined, rowBegin, rowEnd, stream));
    if (timed) {
        MFSR_HIP_TRY(hipEventRecord(b->evStop[b->nEvents], mfsr_s(stream)));
        b->nEvents++;
        b->nFramesTimed += nFrames;
    }
    return MFSR_OK;
}

// Stripe plan of rank `rank` of `worldSize`: its HR rows and the rows of every per-frame product its fuse reads.
// Pure host arithmetic (no device), shared by csrc/dist.cpp and the Python mirror so that both exchange the same rows.
extern "C" int mfsr_dist_stripe_plan(const mfsr_config* cfg, int worldSize, int rank, int rawHalo, mfsr_stripe_plan* out)
{
    MFSR_REQUIRE(out != nullptr && worldSize >= 1 && rank >= 0 && rank < worldSize && rawHalo >= 4);
    TRY(validate(cfg));
    const int s = cfg->scale, W = cfg->width, H = cfg->height;
    const int hrH = s * H;
    const int th = cfg->mono ? H : H / 2, hh = H / 2;
    (void)W;
    // stripes start on multiples of 16 HR rows (tile rows of every fuse kernel, mfsr_accumulateSuperResFullRows)
    auto cut = [&](int g) { return g >= worldSize ? hrH : (int)((long long)hrH * g / worldSize / 16 * 16); };
    const int r0 = cut(rank), r1 = cut(rank + 1);
    memset(out, 0, sizeof(*out));
    out->rowBegin = r0;
    out->rowEnd = r1;
    if (r1 <= r0) return MFSR_OK;  // more ranks than 16-row bands: this rank fuses nothing
    // a field with fh rows is sampled at HR row Y through rows floor((Y + .5) * fh / hrH - .5) and the next one
    auto rows_of = [&](int fh, int pad, int* first, int* count) {
        const int per = hrH / fh;  // HR rows per field row (exact: validate() makes H a multiple of 4)
        int a = r0 / per - 1 - pad, b = (r1 + per - 1) / per + 1 + pad;
        if (a < 0) a = 0;
        if (b > fh) b = fh;
        *first = a;
        *count = b - a;
    };
    rows_of(th, 0, &out->flowRow0, &out->flowRows);
    rows_of(hh, 1, &out->maskRow0, &out->maskRows);  // + the 5x5 footprint: site (Y + py) / s / 2, |py| <= 2
    {
        int a = r0 / s - rawHalo, b = (r1 + s - 1) / s + rawHalo;  // raw row (Y + py + round(s * v)) / s: |v| <= rawHalo - 3
        if (a < 0) a = 0;
        if (b > H) b = H;
        out->rawRow0 = a;
        out->rawRows = b - a;
    }
    out->maxFlowY = (float)(rawHalo - 3);
    return MFSR_OK;
}

extern "C" int mfsr_burst_begin(mfsr_burst* b, mfsr_float3* imgOut, mfsr_float3* totalWeights, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && imgOut && totalWeights);
    TRY(flush_pending(b, stream));  // whatever was still waiting belongs to the previous burst
    b->fresh.has = true;
    b->fresh.imgOut = imgOut;
    b->fresh.totalWeights = totalWeights;
    return MFSR_OK;
}

extern "C" int mfsr_burst_flush(mfsr_burst* b, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b != nullptr);
    return flush_pending(b, stream);
}

extern "C" int mfsr_burst_finish(mfsr_burst* b, const mfsr_float3* imgOut, const mfsr_float3* totalWeights,
                                 mfsr_float3* outImg, uint16_t* out16, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && imgOut && totalWeights && (outImg || out16));
    MFSR_REQUIRE(b->haveRef);
    TRY(flush_pending(b, stream));
    const mfsr_config& c = b->cfg;
    Layout& L = b->L;
    const int pitch = 12 * L.hrW;
    if (c.fused) {
        return mfsr_finishFused(imgOut, totalWeights, pitch, (const mfsr_float3*)L.fallback.ptr, L.fallback.pitch, L.W, L.H,
                                0.0f, 1.0f, 0.0f, 1.0f, outImg, pitch, out16, L.hrW, L.hrH, c.weightThreshold,
                                c.applyGamma, 65535.0f, stream);
    }
    MFSR_REQUIRE(outImg != nullptr);  // the unfused chain needs the float image as its in/out buffer
    TRY(mfsr_resampleFloat3((const mfsr_float3*)L.fallback.ptr, L.fallback.pitch, L.W, L.H, outImg, pitch, L.hrW, L.hrH,
                            0.0f, 1.0f, 0.0f, 1.0f, stream));
    TRY(mfsr_ApplyWeighting(outImg, imgOut, totalWeights, L.hrW, L.hrH, pitch, c.weightThreshold, stream));
    if (c.applyGamma) TRY(mfsr_GammasRGB(outImg, L.hrW, L.hrH, pitch, stream));
    if (out16) TRY(mfsr_quantize(outImg, pitch, out16, nullptr, L.hrW, L.hrH, 65535.0f, stream));
    return MFSR_OK;
}

// finish on the HR row stripe [row0, row0+rows): used after a reduce-scatter of
// the accumulators, where each rank normalises only its own stripe.  Pointers are
// those of the FULL images; only the stripe is read/written.
extern "C" int mfsr_burst_finish_rows(mfsr_burst* b, const mfsr_float3* imgOut, const mfsr_float3* totalWeights,
                                      mfsr_float3* outImg, uint16_t* out16, int row0, int rows, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && imgOut && totalWeights && (outImg || out16));
    MFSR_REQUIRE(b->haveRef);
    const mfsr_config& c = b->cfg;
    Layout& L = b->L;
    MFSR_REQUIRE(row0 >= 0 && rows > 0 && row0 + rows <= L.hrH);
    TRY(flush_pending(b, stream));
    const int pitch = 12 * L.hrW;
    const size_t off = (size_t)row0 * pitch;
    return mfsr_finishFusedRows((const mfsr_float3*)((const char*)imgOut + off),
                                (const mfsr_float3*)((const char*)totalWeights + off), pitch,
                                (const mfsr_float3*)L.fallback.ptr, L.fallback.pitch, L.W, L.H, 0.0f, 1.0f, 0.0f, 1.0f,
                                outImg ? (mfsr_float3*)((char*)outImg + off) : nullptr, pitch,
                                out16 ? out16 + (size_t)row0 * L.hrW * 3 : nullptr, L.hrW, rows, c.weightThreshold,
                                c.applyGamma, 65535.0f, row0, L.hrH, stream);
}

// ---- host-frame bursts (cfg.uploadRing) --------------------------------------------------------------------------
// enqueue the upload of hostRaw into upload slot `us` on the copy stream (after the slot's last consumer) and make the
// compute stream wait for it
static int upload_into(mfsr_burst* b, int us, uint16_t* dst, const uint16_t* hostRaw, mfsr_stream_t stream)
{
    if (b->freeRecorded[us]) {
        MFSR_HIP_TRY(hipStreamWaitEvent(b->copyStream, b->evFree[us], 0));
        b->freeRecorded[us] = false;
    }
    // A 2-D copy, like the download (mfsr_burst_finish_host): measured on this ROCm (tools/pcie_duplex2.py,
    // profiles/r04_pcie_duplex_mechanisms.txt), 1-D uploads queued against 2-D downloads SERIALISE -- 16 uploads + one image:
    // 8.38 ms, the sum of the two directions alone -- while 2-D uploads and 2-D downloads overlap on the full-duplex link:
    // 4.85 ms = the uploads alone.  (A 1-D download would overlap too, but it is a blit kernel whose PCIe-bound stores slow
    // the fuse launches 4-5x.)  MFSR_UPLOAD_1D=1: the round-3 form (A/B).
    static const bool up1d = [] {
        const char* e = getenv("MFSR_UPLOAD_1D");
        return e && e[0] == '1';
    }();
    if (up1d)
        MFSR_HIP_TRY(hipMemcpyAsync(dst, hostRaw, (size_t)b->L.W * b->L.H * 2, hipMemcpyHostToDevice, b->copyStream));
    else
        MFSR_HIP_TRY(hipMemcpy2DAsync(dst, (size_t)b->L.W * 2, hostRaw, (size_t)b->L.W * 2, (size_t)b->L.W * 2, (size_t)b->L.H,
                                      hipMemcpyHostToDevice, b->copyStream));
    MFSR_HIP_TRY(hipEventRecord(b->evUp[us], b->copyStream));
    // the compute stream waits for it when the first kernel that reads an uploaded frame is about to be enqueued
    // (wait_uploads): ONE wait for the last upload of a group instead of one per frame -- a cross-queue wait costs the
    // compute queue ~15 us even when it is already satisfied (four in a row before a group's first kernel: 55-68 us in the
    // trace of back-to-back bursts)
    (void)stream;
    b->upPending = true;
    b->upPendingSlot = us;
    return MFSR_OK;
}

extern "C" int mfsr_burst_set_reference_host(mfsr_burst* b, const uint16_t* hostRaw, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && hostRaw);
    MFSR_REQUIRE(b->copyStream != nullptr);  // cfg.uploadRing > 0
    TRY(flush_pending(b, stream, false));    // a frame still waiting reads the previous reference's slots
    static const bool adaptive = [] {
        const char* e = getenv("MFSR_HOST_ADAPTIVE");
        return !(e && e[0] == '0');
    }();
    b->hostBusy = false;
    if (adaptive && b->downRecorded) {
        // (an event query is not allowed while the caller's stream is being captured into a graph)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(mfsr_s(stream), &cap) == hipSuccess && cap == hipStreamCaptureStatusNone)
            b->hostBusy = hipEventQuery(b->evDown) == hipErrorNotReady;
    }
    const int i = b->refCounter++ & 1;
    const int us = b->cfg.uploadRing + i;
    // The slot's previous reference (two bursts ago) was released by that burst's mfsr_burst_finish_host -- so that this
    // upload does not wait for the tail of the burst before this one, which is still being fused and downloaded when
    // bursts come back to back.  A burst that ended another way: the slot was last read by work already enqueued on the
    // compute stream.
    if (!b->freeRecorded[us]) {
        MFSR_HIP_TRY(hipEventRecord(b->evFree[us], mfsr_s(stream)));
        b->freeRecorded[us] = true;
    }
    b->refSlot = us;
    b->nPrefetched = b->prefetchHead = 0;
    TRY(upload_into(b, us, b->L.refRaw[i], hostRaw, stream));
    b->refHost = hostRaw;
    b->refDev = b->L.refRaw[i];
    return mfsr_burst_set_reference(b, b->refDev, stream);
}

extern "C" int mfsr_burst_add_frame_host(mfsr_burst* b, const uint16_t* hostRaw, int isReference, mfsr_float3* imgOut,
                                         mfsr_float3* totalWeights, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && hostRaw && imgOut && totalWeights);
    MFSR_REQUIRE(b->copyStream != nullptr);
    static const bool banded = [] {
        const char* e = getenv("MFSR_HOST_BANDS");
        return !(e && e[0] == '0');
    }();
    if (isReference && hostRaw == b->refHost && b->refDev)
        return add_frame_impl(b, b->refDev, 1, imgOut, totalWeights, banded, stream);
    if (b->prefetchHead < b->nPrefetched) {
        // announced by mfsr_burst_prefetch_host: the copy is already on the copy stream
        if (b->prefetched[b->prefetchHead].host == hostRaw) {
            const int ps = b->prefetched[b->prefetchHead++].slot;
            b->upPending = true;        // this frame's kernels wait for THIS frame's copy (the copy stream is in order)
            b->upPendingSlot = ps;
            return add_frame_impl(b, b->L.rawRing[ps], isReference, imgOut, totalWeights, banded, stream);
        }
        b->nPrefetched = b->prefetchHead = 0;  // another order than announced: the remaining copies are simply not used
    }
    const int us = b->upCounter++ % b->cfg.uploadRing;
    // a ring shorter than the fuse group: the slot may still hold a frame that waits for the rest of its group
    bool waiting = false, waitingHeld = false;
    for (int i = 0; i < b->pend.n; i++) waiting = waiting || b->pend.raw[i] == b->L.rawRing[us];
    for (int i = 0; b->heldHas && i < b->held.n; i++) waitingHeld = waitingHeld || b->held.raw[i] == b->L.rawRing[us];
    if (waiting)
        TRY(accumulate_pending(b, stream));
    else if (waitingHeld)
        TRY(fuse_held(b, stream));  // only the held group: the frames waiting after it keep their group
    TRY(upload_into(b, us, b->L.rawRing[us], hostRaw, stream));
    return add_frame_impl(b, b->L.rawRing[us], isReference, imgOut, totalWeights, banded, stream);
}

extern "C" int mfsr_burst_prefetch_host(mfsr_burst* b, const uint16_t* const* hostRaws, int nFrames, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && hostRaws && nFrames >= 0);
    MFSR_REQUIRE(b->copyStream != nullptr);
    b->nPrefetched = b->prefetchHead = 0;
    // only at the start of a burst: no frame may be waiting in a slot the copies would overwrite
    if (b->pend.n > 0 || b->heldHas) return MFSR_OK;
    for (int k = 0; k < nFrames && b->nPrefetched < b->cfg.uploadRing; k++) {
        MFSR_REQUIRE(hostRaws[k] != nullptr);
        if (hostRaws[k] == b->refHost && b->refDev) continue;   // add_frame_host re-uses the uploaded reference
        const int us = b->upCounter++ % b->cfg.uploadRing;
        TRY(upload_into(b, us, b->L.rawRing[us], hostRaws[k], stream));
        b->prefetched[b->nPrefetched].host = hostRaws[k];
        b->prefetched[b->nPrefetched].slot = us;
        b->nPrefetched++;
    }
    b->upPending = false;   // (every consumer sets its own wait in add_frame_host)
    return MFSR_OK;
}

// finish + download in row bands.  What is still waiting to be fused (normally the burst's last group, held back by
// add_frame) is fused band by band, each band is normalised as soon as it is complete, and its u16 rows start their way to
// the host while the next band is being fused: the 6 B/px download overlaps the tail of the compute instead of following
// it.  Row-window fuse / finish are bit-identical to the whole-frame launches (the multi-GPU stripes rest on the same).
// The download is a 2-D copy: the runtime hands those to an SDMA engine, whereas hipMemcpyAsync(DeviceToHost) is carried out
// by a blit KERNEL on this ROCm (no memory-copy record in a rocprofv3 trace, a __amd_rocclr_copyBuffer dispatch instead) whose
// PCIe-bound stores slow a warp+fuse launch running beside it 4-5x (profiles/r03_e2e_timeline_blit.txt) -- as did a
// hand-written copy kernel with a 32-workgroup grid: it is the store path that clogs, not the wave slots.
// (One more stream in the context -- a second upload stream was tried -- maps two streams onto one hardware queue and costs
// 4 ms per burst: the context stays at its four streams.)
// every reader of the burst's host reference has been enqueued on `stream`: its upload slot may be refilled after them
static int release_ref_slot(mfsr_burst* b, mfsr_stream_t stream)
{
    if (b->refSlot >= 0) {
        MFSR_HIP_TRY(hipEventRecord(b->evFree[b->refSlot], mfsr_s(stream)));
        b->freeRecorded[b->refSlot] = true;
        b->refSlot = -1;
    }
    return MFSR_OK;
}

extern "C" int mfsr_burst_finish_host(mfsr_burst* b, const mfsr_float3* imgOut, const mfsr_float3* totalWeights,
                                      uint16_t* out16Dev, uint16_t* out16Host, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && imgOut && totalWeights && out16Dev && out16Host);
    MFSR_REQUIRE(b->downStream != nullptr);  // cfg.uploadRing > 0
    MFSR_REQUIRE(b->haveRef);
    const mfsr_config& c = b->cfg;
    Layout& L = b->L;
    // the previous image may still be on its way to the host out of out16Dev
    if (b->downRecorded) MFSR_HIP_TRY(hipStreamWaitEvent(mfsr_s(stream), b->evDown, 0));
    static const int nBandsEnv = [] {
        const char* e = getenv("MFSR_HOST_BANDS");
        return e ? atoi(e) : 8;
    }();
    int nBands = nBandsEnv < 1 ? 1 : (nBandsEnv > 16 ? 16 : nBandsEnv);
    // The bands exist for the LATENCY of one burst (its download starts after 1/8 of the tail).  A burst that was enqueued
    // while the previous one's download was still in flight (b->hostBusy: bursts back to back) gains nothing from them -- its
    // download runs under the next burst's compute anyway -- and pays 16 band-sized fuse launches + 8 finish launches for it:
    // such a burst takes MFSR_HOST_BANDS_BUSY bands (default 2).
    static const int nBandsBusy = [] {
        const char* e = getenv("MFSR_HOST_BANDS_BUSY");
        return e ? atoi(e) : 2;
    }();
    // (only where the image is small against the burst -- x2: 199 MB down for 265 MB up.  At x4 the download, 796 MB, is as
    // long as the burst's compute: it has to start with the first band, or the NEXT burst's finish waits for it to leave
    // out16Dev: 18.1 instead of 17.0 ms per burst with two bands)
    const double outBytes = (double)L.hrW * L.hrH * 6.0, inBytes = (double)L.W * L.H * 2.0 * c.frames;
    if (b->hostBusy && nBandsBusy >= 1 && nBandsBusy < nBands && outBytes <= 1.5 * inBytes) nBands = nBandsBusy;
    bool heldGroup = b->pend.n > 0 && b->pend.imgOut == imgOut && b->pend.totalWeights == totalWeights && c.fused;
    if (b->heldHas && !(heldGroup && b->held.imgOut == imgOut && b->held.totalWeights == totalWeights)) {
        TRY(flush_pending(b, stream));  // a held group without the last one, or other accumulators: the ordinary way
        heldGroup = false;
    }
    if (!heldGroup) TRY(flush_pending(b, stream));  // (another accumulator pair, or the unfused chain: nothing to pipeline)
    const size_t rowBytes = (size_t)L.hrW * 6;
    if (nBands == 1 && !heldGroup) {
        TRY(mfsr_burst_finish(b, imgOut, totalWeights, nullptr, out16Dev, stream));
        MFSR_HIP_TRY(hipEventRecord(b->evFinished, mfsr_s(stream)));
        MFSR_HIP_TRY(hipStreamWaitEvent(b->downStream, b->evFinished, 0));
        MFSR_HIP_TRY(hipMemcpy2DAsync(out16Host, rowBytes, out16Dev, rowBytes, rowBytes, (size_t)L.hrH, hipMemcpyDeviceToHost,
                                      b->downStream));
        MFSR_HIP_TRY(hipEventRecord(b->evDown, b->downStream));
        b->downRecorded = true;
        return release_ref_slot(b, stream);
    }
    mfsr_burst::Pending p = b->pend, p0;
    p0.n = 0;
    int freshNow = 0;
    if (heldGroup) {
        TRY(align_deferred(b, stream));
        p = b->pend;
        b->pend.n = 0;
        if (b->heldHas) {
            p0 = b->held;  // the second-to-last group: fused first on every band (frame order)
            b->heldHas = false;
            b->held.n = 0;
        }
        TRY(join_fuse(b, stream));  // the earlier groups' launches ran on the burst's own stream
        freshNow = b->fresh.has && b->fresh.imgOut == imgOut && b->fresh.totalWeights == totalWeights;
        if (b->fresh.has && !freshNow) {
            const size_t bytes = (size_t)12 * L.hrW * L.hrH;
            MFSR_HIP_TRY(hipMemsetAsync(b->fresh.imgOut, 0, bytes, mfsr_s(stream)));
            MFSR_HIP_TRY(hipMemsetAsync(b->fresh.totalWeights, 0, bytes, mfsr_s(stream)));
        }
        b->fresh.has = false;
    }
    const mfsr_float3 white = {c.white[0], c.white[1], c.white[2]};
    const mfsr_float3 black = {c.black[0], c.black[1], c.black[2]};
    const int tileRows = (L.hrH + 15) / 16;
    if (nBands > tileRows) nBands = tileRows;
    for (int i = 0; i < nBands; i++) {
        const int r0 = (int)((long long)tileRows * i / nBands) * 16;
        int r1 = (int)((long long)tileRows * (i + 1) / nBands) * 16;
        if (r1 > L.hrH || i == nBands - 1) r1 = L.hrH;
        if (r1 <= r0) continue;
        if (heldGroup) {
            const mfsr_burst::Pending* gs[2] = {&p0, &p};
            int fresh = freshNow;
            for (int gi = 0; gi < 2; gi++) {
                const mfsr_burst::Pending& q = *gs[gi];
                if (q.n == 0) continue;
                const mfsr_float4* masks[MFSR_MAX_FUSE_GROUP];
                mfsr_tex2d flows[MFSR_MAX_FUSE_GROUP];
                for (int k = 0; k < q.n; k++) {
                    masks[k] = (const mfsr_float4*)q.mask[k]->ptr;
                    flows[k] = as_tex(*q.flow[k]);
                }
                TRY(mfsr_accumulateSuperResFullRows(q.n, q.raw, const_cast<mfsr_float3*>(imgOut), const_cast<mfsr_float3*>(totalWeights),
                                                    masks, as_tex(L.kparam4), flows, white, black, L.W, L.H, c.scale, 12 * L.hrW,
                                                    q.mask[0]->pitch, fresh, r0, r1, stream));
                fresh = 0;
            }
        }
        const size_t off = (size_t)r0 * 12 * L.hrW;
        TRY(mfsr_finishFusedRows((const mfsr_float3*)((const char*)imgOut + off), (const mfsr_float3*)((const char*)totalWeights + off),
                                 12 * L.hrW, (const mfsr_float3*)L.fallback.ptr, L.fallback.pitch, L.W, L.H, 0.0f, 1.0f, 0.0f, 1.0f,
                                 nullptr, 12 * L.hrW, out16Dev + (size_t)r0 * L.hrW * 3, L.hrW, r1 - r0, c.weightThreshold,
                                 c.applyGamma, 65535.0f, r0, L.hrH, stream));
        if (!b->evBand[i]) MFSR_HIP_TRY(hipEventCreateWithFlags(&b->evBand[i], hipEventDisableTiming));
        MFSR_HIP_TRY(hipEventRecord(b->evBand[i], mfsr_s(stream)));
        MFSR_HIP_TRY(hipStreamWaitEvent(b->downStream, b->evBand[i], 0));
        MFSR_HIP_TRY(hipMemcpy2DAsync((char*)out16Host + (size_t)r0 * rowBytes, rowBytes, (const char*)out16Dev + (size_t)r0 * rowBytes,
                                      rowBytes, rowBytes, (size_t)(r1 - r0), hipMemcpyDeviceToHost, b->downStream));
    }
    if (heldGroup && b->copyStream) {
        // upload slots whose raw frame these launches were the last to read
        for (int gi = 0; gi < 2; gi++) {
            const mfsr_burst::Pending& q = gi ? p : p0;
            for (int j = 0; j < q.n; j++) {
                const int us = upload_slot_of(b, q.raw[j]);
                if (us >= 0) {
                    MFSR_HIP_TRY(hipEventRecord(b->evFree[us], mfsr_s(stream)));
                    b->freeRecorded[us] = true;
                }
            }
        }
    }
    MFSR_HIP_TRY(hipEventRecord(b->evDown, b->downStream));
    b->downRecorded = true;
    return release_ref_slot(b, stream);
}

extern "C" int mfsr_burst_host_sync(mfsr_burst* b)
{
    MFSR_REQUIRE(b != nullptr);
    if (b->downRecorded) MFSR_HIP_TRY(hipEventSynchronize(b->evDown));
    return MFSR_OK;
}

// ---- frame-source plug-in (pull model of multi_frame_sr.cpp:18-49) ------------------------------------------------------
extern "C" int mfsr_burst_process_source(mfsr_burst* b, const mfsr_frame_source* src, mfsr_float3* imgOut, mfsr_float3* totalWeights,
                                         mfsr_float3* outImg, uint16_t* out16, int* framesUsed, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && src && src->next_frame && imgOut && totalWeights && (outImg || out16));
    MFSR_REQUIRE(b->copyStream != nullptr);  // cfg.uploadRing >= 3: the ring slots are the buffers the source fills
    MFSR_REQUIRE(b->cfg.reference == 0);
    if (framesUsed) *framesUsed = 0;
    if (src->reset) src->reset(src->user);
    TRY(mfsr_burst_begin(b, imgOut, totalWeights, stream));
    int n = 0;
    for (; n < b->cfg.frames; n++) {
        // frame 0 goes to a reference slot (it is read again by every later set of reference products), the others
        // through the ring; a slot is handed out once its last consumer has been enqueued
        int us;
        uint16_t* dst;
        if (n == 0) {
            const int i = b->refCounter++ & 1;
            us = b->cfg.uploadRing + i;
            dst = b->L.refRaw[i];
            TRY(flush_pending(b, stream, false));
        } else {
            us = b->upCounter++ % b->cfg.uploadRing;
            dst = b->L.rawRing[us];
            for (int i = 0; i < b->pend.n; i++)  // a ring shorter than the fuse group (see mfsr_burst_add_frame_host)
                if (b->pend.raw[i] == dst) {
                    TRY(accumulate_pending(b, stream));
                    break;
                }
        }
        if (b->freeRecorded[us]) {
            MFSR_HIP_TRY(hipStreamWaitEvent(mfsr_s(stream), b->evFree[us], 0));
            b->freeRecorded[us] = false;
        }
        const int got = src->next_frame(src->user, dst, stream);
        if (got < 0) return got;
        if (got == 0) break;  // exhausted (the reference leaves the output array empty, :42)
        if (n == 0) {
            b->refHost = nullptr;
            b->refDev = dst;
            TRY(mfsr_burst_set_reference(b, dst, stream));
        }
        TRY(mfsr_burst_add_frame(b, dst, n == 0, imgOut, totalWeights, stream));
    }
    if (framesUsed) *framesUsed = n;
    if (n == 0) return MFSR_E_INVALID;  // an empty source has no reference frame
    return mfsr_burst_finish(b, imgOut, totalWeights, outImg, out16, stream);
}

extern "C" int mfsr_burst_debug_views(mfsr_burst* b, mfsr_tex2d* flow, mfsr_tex2d* mask, mfsr_tex2d* kernelParam,
                                      mfsr_tex2d* tracking)
{
    MFSR_REQUIRE(b != nullptr);
    // frame-batched alignment: a frame that is still waiting for its group has no flow / mask yet (flowCur / maskCur would
    // name the previous frame's) -- the caller flushes first (mfsr_burst_flush / finish)
    if ((flow || mask) && b->pend.n > 0 && b->pend.deferred[b->pend.n - 1]) return MFSR_E_INVALID;
    if (flow) *flow = as_tex(*b->flowCur);
    if (mask) *mask = as_tex(*b->maskCur);
    if (kernelParam) *kernelParam = as_tex(b->L.kparam4);
    if (tracking) *tracking = as_tex(b->L.refPyr[0]);
    return MFSR_OK;
}

extern "C" int mfsr_burst_debug_frame_views(mfsr_burst* b, int framesBack, mfsr_tex2d* flow, mfsr_tex2d* mask)
{
    MFSR_REQUIRE(b != nullptr && framesBack >= 0 && framesBack < kRing && framesBack < b->frameCounter);
    const int slot = (int)((b->frameCounter - 1 - framesBack) % kRing);
    MFSR_REQUIRE(b->slotFlow[slot] != nullptr);
    if (flow) *flow = as_tex(*b->slotFlow[slot]);
    if (mask) *mask = as_tex(b->L.maskBuf[slot]);
    return MFSR_OK;
}

extern "C" int mfsr_burst_prealign_result(mfsr_burst* b, mfsr_prealign* hostOut, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && hostOut);
    MFSR_REQUIRE(b->cfg.preAlign && b->L.preResult);
    TRY(align_deferred(b, stream));  // the last frame may still be waiting for its group: its estimate does not exist yet
    MFSR_HIP_TRY(hipMemcpyAsync(hostOut, b->preCur, sizeof(*hostOut), hipMemcpyDeviceToHost, mfsr_s(stream)));
    MFSR_HIP_TRY(hipStreamSynchronize(mfsr_s(stream)));
    return MFSR_OK;
}

// ---- frame streams: sliding window of 2R+1 frames around every frame (the reference's setTemporalAreaRadius(1),
//      finalProject/Project/multi_frame_sr.cpp:182; BASELINE configs[4]) ---------------------------------------------------
// Output t fuses frames [t-R, t+R] (clipped to the stream) with frame t as the reference: what one mfsr_burst_* burst per
// window computes.  A frame takes part in up to 2R+1 windows; its upload and its per-frame products that do not depend
// on the reference (A1 half-resolution RGB, tracking pyramid, pre-alignment search pyramid) are made once, when it
// arrives, and kept in a ring of 2R+1 entries; the burst context gets their descriptors swapped in.
namespace {
struct FrameProducts {
    uint16_t* raw;
    Img half;
    Img pyr[8];
    void* prePyr;
};
struct StreamLayout {
    size_t burstWs, accBytes, entryBytes;
    size_t offBurst, offImg, offTw, offEntries, total;
};
size_t stream_entry(const mfsr_config* c, char* base, FrameProducts* fp)
{
    Layout L;
    make_layout(c, nullptr, &L);
    Bump b{base, 0};
    const int nl = ilog2(L.maxFactor) + 1;
    FrameProducts tmp;
    memset((void*)&tmp, 0, sizeof(tmp));
    tmp.raw = (uint16_t*)b.take((size_t)L.W * L.H * 2);
    tmp.half = b.image(L.hw, L.hh, 12);
    for (int i = 0; i < nl; i++) tmp.pyr[i] = b.image(L.tw >> i, L.th >> i, 4);
    tmp.prePyr = c->preAlign ? b.take(mfsr_preAlign_pyramid_bytes(L.tw, L.th)) : nullptr;
    if (fp) *fp = tmp;
    return align_up(b.off, 256);
}
int stream_layout(const mfsr_config* c, int radius, StreamLayout* S)
{
    memset(S, 0, sizeof(*S));
    Layout L;
    make_layout(c, nullptr, &L);
    S->burstWs = L.total;
    S->accBytes = (size_t)12 * L.hrW * L.hrH;
    S->entryBytes = stream_entry(c, nullptr, nullptr);
    size_t off = 0;
    S->offBurst = off;
    off = align_up(off + S->burstWs, 256);
    S->offImg = off;
    off = align_up(off + S->accBytes, 256);
    S->offTw = off;
    off = align_up(off + S->accBytes, 256);
    S->offEntries = off;
    off += S->entryBytes * (size_t)(2 * radius + 1);
    S->total = align_up(off, 256);
    return MFSR_OK;
}
}  // namespace

struct mfsr_stream {
    mfsr_config cfg;
    int R, cap, hostFrames;
    mfsr_burst* b;
    StreamLayout S;
    mfsr_float3 *imgOut, *totalWeights;
    std::vector<FrameProducts>* fp;
    long long pushed, produced;
    // frames in host memory: uploads on a stream of their own, ordered against the windows that read the slot
    hipStream_t copyStream;
    hipEvent_t evUp, evWindow[64];   // evWindow[j % 64]: output j's work enqueued on the compute stream
    bool windowRecorded[64];
    long long lastWindow;            // index of the last output whose evWindow was recorded (-1: none)
};

extern "C" size_t mfsr_stream_workspace_bytes(const mfsr_config* cfg, int radius)
{
    if (validate(cfg) != MFSR_OK || radius < 0 || radius > 15) return 0;
    StreamLayout S;
    stream_layout(cfg, radius, &S);
    return S.total;
}

extern "C" int mfsr_stream_create(mfsr_stream** out, const mfsr_config* cfg, int radius, int framesInHostMemory, void* workspace,
                                  size_t workspaceBytes)
{
    MFSR_REQUIRE(out && workspace && radius >= 0 && radius <= 15 && ((uintptr_t)workspace & 255) == 0);
    TRY(validate(cfg));
    mfsr_stream* s = new (std::nothrow) mfsr_stream;
    MFSR_REQUIRE(s != nullptr);
    memset((void*)s, 0, sizeof(*s));
    s->cfg = *cfg;
    s->cfg.frames = 2 * radius + 1;
    s->cfg.uploadRing = 0;
    s->R = radius;
    s->cap = 2 * radius + 1;
    s->hostFrames = framesInHostMemory ? 1 : 0;
    stream_layout(&s->cfg, radius, &s->S);
    if (s->S.total > workspaceBytes) {
        fprintf(stderr, "mfsr: stream workspace too small: need %zu bytes, got %zu\n", s->S.total, workspaceBytes);
        delete s;
        return MFSR_E_WORKSPACE;
    }
    char* base = (char*)workspace;
    int rc = mfsr_burst_create(&s->b, &s->cfg, base + s->S.offBurst, s->S.burstWs);
    if (rc != MFSR_OK) {
        delete s;
        return rc;
    }
    s->imgOut = (mfsr_float3*)(base + s->S.offImg);
    s->totalWeights = (mfsr_float3*)(base + s->S.offTw);
    s->lastWindow = -1;
    s->fp = new (std::nothrow) std::vector<FrameProducts>(s->cap);
    if (!s->fp) {
        mfsr_stream_destroy(s);
        return MFSR_E_INVALID;
    }
    for (int i = 0; i < s->cap; i++) stream_entry(&s->cfg, base + s->S.offEntries + s->S.entryBytes * (size_t)i, &(*s->fp)[i]);
    if (s->hostFrames) {
        hipError_t e = hipStreamCreateWithFlags(&s->copyStream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->evUp, hipEventDisableTiming);
        for (int i = 0; i < 64 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&s->evWindow[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            mfsr_stream_destroy(s);
            return MFSR_E_NODEVICE;
        }
    }
    *out = s;
    return MFSR_OK;
}

extern "C" void mfsr_stream_destroy(mfsr_stream* s)
{
    if (!s) return;
    if (s->copyStream) (void)hipStreamSynchronize(s->copyStream);
    if (s->evUp) (void)hipEventDestroy(s->evUp);
    for (int i = 0; i < 64; i++)
        if (s->evWindow[i]) (void)hipEventDestroy(s->evWindow[i]);
    if (s->copyStream) (void)hipStreamDestroy(s->copyStream);
    mfsr_burst_destroy(s->b);
    delete s->fp;
    delete s;
}

// output j from the cached frames [lo, hi]
static int stream_window(mfsr_stream* s, long long j, long long lo, long long hi, mfsr_float3* outImg, uint16_t* out16,
                         mfsr_stream_t stream)
{
    mfsr_burst* b = s->b;
    Layout& L = b->L;
    const int nl = ilog2(L.maxFactor) + 1;
    auto use = [&](const FrameProducts& f, bool asRef) {
        (asRef ? L.refHalf : L.movHalf) = f.half;
        for (int i = 0; i < nl; i++) (asRef ? L.refPyr : L.movPyr)[i] = f.pyr[i];
        (asRef ? L.preRefPyr : L.preMovPyr) = f.prePyr;
    };
    const FrameProducts& ref = (*s->fp)[j % s->cap];
    TRY(mfsr_burst_begin(b, s->imgOut, s->totalWeights, stream));
    use(ref, true);
    b->refPrepared = true;
    int rc = mfsr_burst_set_reference(b, ref.raw, stream);
    b->refPrepared = false;
    if (rc) return rc;
    for (long long k = lo; k <= hi && rc == MFSR_OK; k++) {
        const FrameProducts& f = (*s->fp)[k % s->cap];
        use(f, false);
        b->movPrepared = true;
        rc = mfsr_burst_add_frame(b, f.raw, k == j, s->imgOut, s->totalWeights, stream);
        b->movPrepared = false;
    }
    if (rc) return rc;
    TRY(mfsr_burst_finish(b, s->imgOut, s->totalWeights, outImg, out16, stream));
    if (s->copyStream) {
        MFSR_HIP_TRY(hipEventRecord(s->evWindow[j % 64], mfsr_s(stream)));
        s->windowRecorded[j % 64] = true;
        s->lastWindow = j;
    }
    return MFSR_OK;
}

extern "C" int mfsr_stream_push(mfsr_stream* s, const uint16_t* frame, mfsr_float3* outImg, uint16_t* out16, long long* produced,
                                mfsr_stream_t stream)
{
    MFSR_REQUIRE(s && frame && produced);
    *produced = -1;
    mfsr_burst* b = s->b;
    Layout& L = b->L;
    const long long t = s->pushed;
    FrameProducts& f = (*s->fp)[t % s->cap];
    const size_t rawBytes = (size_t)L.W * L.H * 2;
    if (s->hostFrames) {
        // the slot held frame t - cap, last read by output t - cap + R = t - R - 1
        const long long last = t - s->R - 1;
        if (last >= 0 && s->windowRecorded[last % 64]) MFSR_HIP_TRY(hipStreamWaitEvent(s->copyStream, s->evWindow[last % 64], 0));
        // (2-D, like mfsr_burst_add_frame_host's uploads: overlaps with the 2-D downloads of a caller's results)
        MFSR_HIP_TRY(hipMemcpy2DAsync(f.raw, (size_t)L.W * 2, frame, (size_t)L.W * 2, (size_t)L.W * 2, (size_t)L.H, hipMemcpyHostToDevice,
                                      s->copyStream));
        MFSR_HIP_TRY(hipEventRecord(s->evUp, s->copyStream));
        MFSR_HIP_TRY(hipStreamWaitEvent(mfsr_s(stream), s->evUp, 0));
    } else {
        MFSR_HIP_TRY(hipMemcpyAsync(f.raw, frame, rawBytes, hipMemcpyDeviceToDevice, mfsr_s(stream)));
    }
    // per-frame products, once per frame
    TRY(mfsr_set_cfa_pattern(s->cfg.cfa));
    TRY(prepare_frame(b, f.raw, f.half, f.pyr, stream));
    if (s->cfg.preAlign)
        TRY(mfsr_preAlignPyramid((const float*)f.pyr[0].ptr, L.tw, L.th, f.pyr[0].pitch, f.prePyr, stream));
    s->pushed = t + 1;
    if (t < s->R) return MFSR_OK;
    MFSR_REQUIRE(outImg || out16);
    const long long j = t - s->R;
    TRY(stream_window(s, j, j - s->R > 0 ? j - s->R : 0, t, outImg, out16, stream));
    s->produced = j + 1;
    *produced = j;
    return MFSR_OK;
}

extern "C" int mfsr_stream_drain(mfsr_stream* s, mfsr_float3* outImg, uint16_t* out16, long long* produced, mfsr_stream_t stream)
{
    MFSR_REQUIRE(s && produced && (outImg || out16));
    *produced = -1;
    if (s->produced >= s->pushed) return MFSR_OK;
    const long long j = s->produced;
    TRY(stream_window(s, j, j - s->R > 0 ? j - s->R : 0, s->pushed - 1, outImg, out16, stream));
    s->produced = j + 1;
    *produced = j;
    return MFSR_OK;
}

extern "C" int mfsr_stream_reset(mfsr_stream* s)
{
    MFSR_REQUIRE(s != nullptr);
    // host-frame mode: the next push uploads into slot 0 on the copy stream, which the windows of the stream that is being
    // abandoned may still read on the compute stream: order the copy stream after the last of them (they were enqueued in
    // order), then forget their events -- index j of the new stream must not wait for output j of the old one
    if (s->copyStream && s->lastWindow >= 0 && s->windowRecorded[s->lastWindow % 64])
        MFSR_HIP_TRY(hipStreamWaitEvent(s->copyStream, s->evWindow[s->lastWindow % 64], 0));
    for (int i = 0; i < 64; i++) s->windowRecorded[i] = false;
    s->lastWindow = -1;
    s->pushed = s->produced = 0;
    return MFSR_OK;
}

// ---- whole burst with the joint shift minimiser (stage C of SURVEY.md section 3.3; ShiftMinimizerKernels.cu) ----------------
// Besides every (reference, k) pair the tile tracker also measures every neighbouring pair (k, k+1); per tile the
// N-1 frame-to-frame shifts are the least-squares solution of those m measurements (design matrix: a contiguous run of
// ones per pair, ShiftMinimizerKernels.cu:137), pairs whose residual exceeds 1 px^2 are dropped one at a time and the
// system is solved again (checkForOutliers :81-139), and the reference->k tile shifts are the signed prefix sums
// (getOptimalShifts :179-218).  Those shifts then replace the tracker's in the per-frame chain D (flow field + LK) -> F -> G.
// One fixed launch sequence, no host round trip: the solve/reject loop runs per tile inside mfsr_minimizeShiftsFused.
namespace {
struct JointLayout {
    int N, m, n1, tiles, tcx, tcy;
    size_t entryBytes;
    size_t offEntries, offPairShifts, offRefSq, offMatrix, offMeasured, offOneToOne, offOptimT, offStatus, offInfo, offPtrs, offPitches,
        offOptimal, total;
    int pairPitch;
    size_t pairBytes, refSqBytes;
};
int joint_pairs(int N, int ref, int (*pairs)[2])
{
    int m = 0;
    for (int k = 0; k + 1 < N; k++) {  // neighbours first
        if (pairs) {
            pairs[m][0] = k;
            pairs[m][1] = k + 1;
        }
        m++;
    }
    for (int k = 0; k < N; k++) {  // then the reference against every frame that is not its neighbour
        const int a = k < ref ? k : ref, bb = k < ref ? ref : k;
        if (bb - a < 2) continue;
        if (pairs) {
            pairs[m][0] = a;
            pairs[m][1] = bb;
        }
        m++;
    }
    return m;
}
void joint_layout(const mfsr_config* c, JointLayout* J)
{
    memset(J, 0, sizeof(*J));
    Layout L;
    make_layout(c, nullptr, &L);
    const int last = c->levels - 1;
    J->N = c->frames;
    J->n1 = c->frames - 1;
    J->m = joint_pairs(c->frames, c->reference, nullptr);
    J->tcx = L.tcx[last];
    J->tcy = L.tcy[last];
    J->tiles = J->tcx * J->tcy;
    J->entryBytes = stream_entry(c, nullptr, nullptr);
    J->pairPitch = (int)align_up((size_t)J->tcx * 8, 64);
    J->pairBytes = align_up((size_t)J->pairPitch * J->tcy, 256);
    size_t refSq = 0;
    for (int l = 0; l < c->levels; l++) refSq += align_up((size_t)L.tcx[l] * L.tcy[l] * 4, 256);
    J->refSqBytes = refSq;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        off = align_up(off, 256);
        const size_t o = off;
        off += bytes;
        return o;
    };
    J->offEntries = take(J->entryBytes * (size_t)J->N);
    J->offPairShifts = take(J->pairBytes * (size_t)J->m);
    J->offRefSq = take(J->refSqBytes * (size_t)J->N);
    J->offMatrix = take(sizeof(float) * (size_t)J->tiles * J->n1 * J->m);
    J->offMeasured = take(8 * (size_t)J->tiles * J->m);
    J->offOneToOne = take(8 * (size_t)J->tiles * J->n1);
    J->offOptimT = take(sizeof(float) * 2 * (size_t)J->tiles * J->m);
    J->offStatus = take(sizeof(int) * (size_t)J->tiles);
    J->offInfo = take(sizeof(int) * (size_t)J->tiles);
    J->offPtrs = take(sizeof(void*) * (size_t)J->m);
    J->offPitches = take(sizeof(int) * (size_t)J->m);
    J->offOptimal = take(J->pairBytes);
    J->total = align_up(off, 256);
}
}  // namespace

extern "C" size_t mfsr_burst_joint_workspace_bytes(const mfsr_config* cfg)
{
    if (validate(cfg) != MFSR_OK || cfg->frames < 2 || cfg->frames - 1 > 63) return 0;
    JointLayout J;
    joint_layout(cfg, &J);
    return J.total;
}

extern "C" int mfsr_burst_process_joint(mfsr_burst* b, const uint16_t* const* frames, void* jointWorkspace, size_t jointBytes,
                                        mfsr_float3* imgOut, mfsr_float3* totalWeights, mfsr_stream_t stream)
{
    MFSR_REQUIRE(b && frames && jointWorkspace && imgOut && totalWeights && ((uintptr_t)jointWorkspace & 255) == 0);
    const mfsr_config& c = b->cfg;
    Layout& L = b->L;
    MFSR_REQUIRE(c.frames >= 2 && c.frames - 1 <= 63);
    if (c.preAlign || !c.fused) {
        fprintf(stderr, "mfsr: the joint mode needs cfg.fused = 1 and cfg.preAlign = 0 (the minimiser sums tracked shifts only)\n");
        return MFSR_E_UNSUPPORTED;
    }
    JointLayout J;
    joint_layout(&c, &J);
    if (J.total > jointBytes) {
        fprintf(stderr, "mfsr: joint workspace too small: need %zu bytes, got %zu\n", J.total, jointBytes);
        return MFSR_E_WORKSPACE;
    }
    for (int k = 0; k < c.frames; k++) MFSR_REQUIRE(frames[k] != nullptr);
    char* base = (char*)jointWorkspace;
    const int N = c.frames, last = c.levels - 1, nl = ilog2(L.maxFactor) + 1;
    std::vector<FrameProducts> fp(N);
    TRY(mfsr_set_cfa_pattern(c.cfa));
    TRY(flush_pending(b, stream, false));
    // per-frame products of every frame (each serves as reference and as moved frame of some pair)
    for (int k = 0; k < N; k++) {
        stream_entry(&c, base + J.offEntries + J.entryBytes * (size_t)k, &fp[k]);
        fp[k].raw = const_cast<uint16_t*>(frames[k]);  // read only
        TRY(prepare_frame(b, frames[k], fp[k].half, fp[k].pyr, stream));
    }
    auto use = [&](const FrameProducts& f, bool asRef) {
        (asRef ? L.refHalf : L.movHalf) = f.half;
        for (int i = 0; i < nl; i++) (asRef ? L.refPyr : L.movPyr)[i] = f.pyr[i];
    };
    const Img savedRefHalf = L.refHalf, savedMovHalf = L.movHalf;
    Img savedRefPyr[8], savedMovPyr[8];
    float* savedRefSq[kMaxLevels];
    for (int i = 0; i < 8; i++) {
        savedRefPyr[i] = L.refPyr[i];
        savedMovPyr[i] = L.movPyr[i];
    }
    for (int l = 0; l < kMaxLevels; l++) savedRefSq[l] = L.refSq[l];
    auto restore = [&]() {
        L.refHalf = savedRefHalf;
        L.movHalf = savedMovHalf;
        for (int i = 0; i < 8; i++) {
            L.refPyr[i] = savedRefPyr[i];
            L.movPyr[i] = savedMovPyr[i];
        }
        for (int l = 0; l < kMaxLevels; l++) L.refSq[l] = savedRefSq[l];
        b->refPrepared = b->movPrepared = false;
        b->givenShifts = nullptr;
        b->refStale = true;  // L.refHalf / L.refPyr no longer belong to the reference set_reference saw
    };
    struct Guard {
        decltype(restore)& r;
        ~Guard() { r(); }
    } guard{restore};

    // sum(ref^2) tables of every frame that is the first of a pair
    int pairs[2 * 64][2];
    const int m = joint_pairs(N, c.reference, pairs);
    auto refsq_of = [&](int k, int l) {
        size_t o = 0;
        for (int i = 0; i < l; i++) o += align_up((size_t)L.tcx[i] * L.tcy[i] * 4, 256);
        return (float*)(base + J.offRefSq + J.refSqBytes * (size_t)k + o);
    };
    std::vector<char> haveSq(N, 0);
    // B: every pair through the coarse -> fine tracker
    const float2* hostPtrs[2 * 64];
    int hostPitches[2 * 64];
    for (int p = 0; p < m; p++) {
        const int a = pairs[p][0], bb = pairs[p][1];
        use(fp[a], true);
        use(fp[bb], false);
        for (int l = 0; l < c.levels; l++) {
            L.refSq[l] = refsq_of(a, l);
            if (!haveSq[a]) {
                const Img& ref = L.refPyr[ilog2(c.levelFactor[l])];
                TRY(mfsr_tileSquaredSums((const float*)ref.ptr, L.refSq[l], ref.w, ref.h, ref.pitch, c.maxShift[l], c.tileSize[l],
                                         L.tcx[l], L.tcy[l], stream));
            }
        }
        haveSq[a] = 1;
        TRY(track_tiles(b, nullptr, stream));
        char* dst = base + J.offPairShifts + J.pairBytes * (size_t)p;
        MFSR_HIP_TRY(hipMemcpy2DAsync(dst, J.pairPitch, L.shifts[last].ptr, L.shifts[last].pitch, (size_t)J.tcx * 8, J.tcy,
                                      hipMemcpyDeviceToDevice, mfsr_s(stream)));
        hostPtrs[p] = (const float2*)dst;
        hostPitches[p] = J.pairPitch;
    }
    // C: [tile][m] measurements, design matrix, per-tile solve / reject loop, prefix sums
    b->jointHost.assign((size_t)J.n1 * m + 2 * (size_t)m * sizeof(void*) / sizeof(float) + 2 * m + 8, 0.0f);
    float* hA = b->jointHost.data();  // column-major m x n1 (ShiftMinimizerKernels.cu:137): row p has ones in columns a..b-1
    for (int p = 0; p < m; p++)
        for (int col = pairs[p][0]; col < pairs[p][1]; col++) hA[p + (size_t)col * m] = 1.0f;
    char* hTail = (char*)(hA + (size_t)J.n1 * m);
    hTail = (char*)(((uintptr_t)hTail + 15) & ~(uintptr_t)15);
    memcpy(hTail, hostPtrs, sizeof(void*) * m);
    memcpy(hTail + sizeof(void*) * m, hostPitches, sizeof(int) * m);
    float* dA = (float*)(base + J.offMatrix);
    MFSR_HIP_TRY(hipMemcpyAsync(dA, hA, sizeof(float) * (size_t)J.n1 * m, hipMemcpyHostToDevice, mfsr_s(stream)));
    MFSR_HIP_TRY(hipMemcpyAsync(base + J.offPtrs, hTail, sizeof(void*) * m, hipMemcpyHostToDevice, mfsr_s(stream)));
    MFSR_HIP_TRY(hipMemcpyAsync(base + J.offPitches, hTail + sizeof(void*) * m, sizeof(int) * m, hipMemcpyHostToDevice, mfsr_s(stream)));
    TRY(mfsr_copyShiftMatrix(dA, J.tiles, N, m, stream));
    mfsr_float2* measured = (mfsr_float2*)(base + J.offMeasured);
    TRY(mfsr_concatenateShifts((const mfsr_float2* const*)(base + J.offPtrs), (int*)(base + J.offPitches), measured, m, J.tcx, J.tcy,
                               stream));
    mfsr_float2* oneToOne = (mfsr_float2*)(base + J.offOneToOne);
    TRY(mfsr_minimizeShiftsFused(dA, measured, oneToOne, (float*)(base + J.offOptimT), (int*)(base + J.offStatus),
                                 (int*)(base + J.offInfo), J.tiles, N, m, stream));
    // D, F, G per frame with the minimiser's tile shifts
    use(fp[c.reference], true);
    for (int l = 0; l < c.levels; l++) L.refSq[l] = refsq_of(c.reference, l);  // (unused by the chain below; kept consistent)
    b->refPrepared = true;
    TRY(mfsr_burst_begin(b, imgOut, totalWeights, stream));
    TRY(mfsr_burst_set_reference(b, frames[c.reference], stream));
    // set_reference recomputed the reference's tile sums into L.refSq: harmless
    Img optimal;
    optimal.ptr = base + J.offOptimal;
    optimal.pitch = J.pairPitch;
    optimal.w = J.tcx;
    optimal.h = J.tcy;
    for (int k = 0; k < N; k++) {
        if (k != c.reference) {
            TRY(mfsr_getOptimalShifts((mfsr_float2*)optimal.ptr, oneToOne, N, J.tcx, J.tcy, optimal.pitch, c.reference, k, stream));
            use(fp[k], false);
            b->movPrepared = true;
            b->givenShifts = &optimal;
        }
        const int rc = mfsr_burst_add_frame(b, frames[k], k == c.reference, imgOut, totalWeights, stream);
        b->movPrepared = false;
        b->givenShifts = nullptr;
        if (rc) return rc;
    }
    TRY(mfsr_burst_flush(b, stream));
    return MFSR_OK;
}

// ---- C: joint shift minimiser driver (solve -> checkForOutliers until every tile
//      reports -1).  status/inversionInfo are device arrays of tileCount ints; the
//      loop is bounded by shiftCount rounds and polls the device once per round. ----
extern "C" int mfsr_minimizeShifts(float* shiftMatrix, mfsr_float2* measuredShifts, mfsr_float2* shiftsOneToOne,
                                   float* optimShiftsT, int* status, int* inversionInfo, int tileCount, int imageCount,
                                   int shiftCount, int* roundsOut, mfsr_stream_t stream)
{
    MFSR_REQUIRE(shiftMatrix && measuredShifts && shiftsOneToOne && optimShiftsT && status && inversionInfo);
    MFSR_REQUIRE(tileCount > 0 && imageCount > 1 && shiftCount > 0);
    MFSR_HIP_TRY(hipMemsetAsync(status, 0, sizeof(int) * (size_t)tileCount, mfsr_s(stream)));
    int* hstatus = new (std::nothrow) int[tileCount];
    MFSR_REQUIRE(hstatus != nullptr);
    int rounds = 0, rc = MFSR_OK;
    for (; rounds < shiftCount + 1; rounds++) {
        rc = mfsr_solveShiftsBatched(shiftMatrix, measuredShifts, shiftsOneToOne, optimShiftsT, inversionInfo, tileCount,
                                     imageCount, shiftCount, stream);
        if (rc) break;
        rc = mfsr_checkForOutliers(measuredShifts, optimShiftsT, shiftMatrix, status, inversionInfo, tileCount, imageCount,
                                   shiftCount, stream);
        if (rc) break;
        hipError_t e = hipMemcpyAsync(hstatus, status, sizeof(int) * (size_t)tileCount, hipMemcpyDeviceToHost, mfsr_s(stream));
        if (e == hipSuccess) e = hipStreamSynchronize(mfsr_s(stream));
        if (e != hipSuccess) {
            rc = (int)e;
            break;
        }
        bool done = true;
        for (int i = 0; i < tileCount; i++)
            if (hstatus[i] >= 0) {
                done = false;
                break;
            }
        if (done) {
            rounds++;
            break;
        }
    }
    delete[] hstatus;
    if (roundsOut) *roundsOut = rounds;
    return rc;
}
