// abi.cpp — implementation of include/crt_abi.h: context, scene flattening + upload, launches, read-back.
// Host-only logic; the kernels live in device/kernels.hip.  There is NO CPU rendering path in this library:
// without a HIP device crt_create fails with CRT_ERR_NO_DEVICE.
#include "../../include/crt_abi.h"
#include "device/layout.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <algorithm>
#include <vector>
#include <chrono>

extern "C" uint32_t crt_render_resident_waves(int, int, uint32_t);
extern "C" uint32_t crt_probe_paths();
extern "C" hipError_t crt_launch_render(const crt::Scene*, void*, crt::Counters*, unsigned long long*, const uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, int, const uint32_t*, uint32_t, uint32_t*, uint32_t, unsigned long long*, hipStream_t);
extern "C" size_t crt_pool_scratch_bytes_per_window(uint32_t);
extern "C" hipError_t crt_launch_render_pool(const crt::Scene*, void*, void*, crt::Counters*, unsigned long long*, const uint32_t*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, int, uint32_t, uint32_t*, unsigned long long*, hipStream_t);
extern "C" uint32_t crt_pool_streams(uint32_t frames);
extern "C" uint32_t crt_narrow_max_lanes(void);
extern "C" hipError_t crt_launch_probe(const crt::Scene*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*, hipStream_t);
extern "C" hipError_t crt_launch_render_narrow(const crt::Scene*, void*, crt::Counters*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t*, uint32_t, uint32_t*, hipStream_t);
extern "C" hipError_t crt_launch_accumulate(const void*, void*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, hipStream_t);
extern "C" hipError_t crt_launch_find_nearest(const crt::Scene*, const void*, void*, uint32_t, crt::Counters*, uint32_t, uint32_t*, hipStream_t);
namespace crt { struct AltAccelDev; }
extern "C" hipError_t crt_launch_whitted(const crt::Scene*, int, const crt::AltAccelDev*, void*, uint32_t*, crt::Counters*, uint32_t, hipStream_t);
extern "C" hipError_t crt_launch_render_alt(int, const crt::Scene*, const crt::AltAccelDev*, void*, crt::Counters*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, hipStream_t);
extern "C" hipError_t crt_launch_resolve(const void*, uint32_t*, float*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, float, hipStream_t);

static_assert(sizeof(crt_bvh_node) == 32 && sizeof(crt_tri) == 112 && sizeof(crt_tlas_node) == 32, "reference layouts");
static_assert(sizeof(crt::NodePair) == 64 && sizeof(crt::LeafTri) == 48 && sizeof(crt::ShadeTri) == 64 && sizeof(crt::TlasNode) == 32 &&
              sizeof(crt::Instance) == 128 && sizeof(crt::Material) == 32 && offsetof(crt::Instance, T) == 64, "device layouts");
static_assert(sizeof(crt_counters) == sizeof(crt::Counters), "counter layout");
static_assert(sizeof(crt_kd_node) == 48, "flat KD node");
// device view of the alternative accelerators (device/alt_accel.hip)
namespace crt {
struct KdNode; struct AltTri;
struct AltAccelDev {
    const void* kdNodes; const uint32_t* kdRefs; uint32_t kdStack;
    const void* tris;
    int32_t res[3]; float cell[3]; float lo[3], hi[3]; const uint32_t* cellStart; const int32_t* cellRefs;
};
}
namespace crt {
struct PrimDev {                          // = device/render_prim.hip
    float quadInvT[12], quadNrm[3], quadSize; float spherePos[3], pad0; float cubeInvM[12], cubeM[12], cubeMin[3], cubeMax[3];
    float torusInvT[12], torusT[12], rt2, rc2, r2, pad1; float refl[11], refr[11], absorb[33]; float pad2; const uint32_t* red; const uint32_t* blue;
};
}
extern "C" hipError_t crt_launch_find_nearest_prim(const crt::PrimDev*, const void*, void*, uint32_t, hipStream_t);
extern "C" hipError_t crt_launch_render_prim(const crt::Scene*, const crt::PrimDev*, void*, crt::Counters*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, hipStream_t);
extern "C" hipError_t crt_launch_check_reciprocals(unsigned long long*, hipStream_t);
extern "C" hipError_t crt_launch_find_nearest_alt(int, const crt::Scene*, const crt::AltAccelDev*, const void*, void*, uint32_t, uint32_t*, hipStream_t);

namespace {

thread_local std::string g_createError;

// Test / diagnostic switches (CRT_RENDER_KERNEL, CRT_LAT_*, CRT_SPLIT_*, CRT_DEBUG_*, CRT_NARROW_*, CRT_PLAN_*) are environment variables that are looked at
// ONLY after the process has called crt_debug_enable_hooks(1) — the test suite and the tools do, a host application never does, so a variable that happens to be
// exported in its environment cannot change what the library does.
bool g_hooks = false;
inline const char* hook(const char* name) { return g_hooks ? getenv(name) : nullptr; }

struct EventPair { hipEvent_t a, b; int mode = -1; bool seen = false; };   // mode: latency-mode tag of a single-window launch (0 wide, 1 narrow), -1 otherwise

} // namespace

struct crt_ctx {
    crt_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;
    int tilesX = 0, tilesY = 0;
    uint32_t tileFirst = 0, tileStride = 1, tileCount = 0;
    // device memory
    void* dAccOwned = nullptr; void* dAcc = nullptr;
    // Render launches (independent: different spp windows) rotate over `streams` so that consecutive crt_render calls overlap on
    // the GPU; the ordered accumulate kernels run on the main stream behind events.  Their sample slabs are regions of ONE pool
    // handed out as a ring: launch order = accumulate order = release order, so the oldest region is always the next to free.
    std::vector<hipStream_t> streams; uint64_t launchSeq = 0;
    hipStream_t narrowStream = nullptr;                          // highest priority: render_narrow_kernel's workgroups (32+ KB of LDS each) must not queue behind the wide kernel's wavefronts
    struct Region { size_t off, bytes; hipEvent_t freed; };      // freed: recorded on the main stream behind the region's accumulate
    std::deque<Region> inflight;
    char* pool = nullptr; size_t poolBytes = 0, poolHead = 0; bool poolCapped = false;   // capped: already as large as the HBM budget allows
    hipEvent_t mustWait = nullptr;                                // `freed` of the newest region whose space was handed out again
    std::vector<hipEvent_t> freeEvents;
    std::vector<hipEvent_t> doneEvents;
    crt::Scene hScene{};
    crt::Counters* dCounters = nullptr;
    uint32_t* dQueryCursor = nullptr;      // the ray cursor of the persistent query kernels (zeroed by each launch)
    uint32_t* dPixels = nullptr; float* dTileSums = nullptr;
    unsigned long long* dTileClocks = nullptr;
    std::vector<void*> sceneAllocs;
    // dispatch-order heuristic: owned tiles that can see the scene's meshes come first (see render_tiles_kernel)
    float meshLo[3] = {0, 0, 0}, meshHi[3] = {0, 0, 0}; bool orderDirty = true;
    uint32_t* dTileOrder = nullptr;
    uint32_t* hTileOrder[2] = {nullptr, nullptr}; hipEvent_t orderCopied[2] = {nullptr, nullptr}; hipEvent_t orderReady = nullptr; int orderFlip = 0;
    bool haveScene = false;
    // what crt_update_scene needs of the last upload: a host mirror of the geometry buffer and where each BVH's records start
    struct Flat { uint64_t topOff = 0; uint32_t topCount = 0, rootRef = 0; int32_t kind = 0; uint64_t leafOff = 0, tlasOff = 0, tlasPairOff = 0, instOff = 0, shadeOff = 0; uint32_t tlasNodeCount = 0, maxHeight = 0;
                  std::vector<uint64_t> pairBase, triBase; std::vector<uint32_t> nodesUsed, triCount; std::vector<char> geom; } flat;
    char* hStage[2] = {nullptr, nullptr}; size_t stageBytes[2] = {0, 0}; hipEvent_t stageCopied[2] = {nullptr, nullptr}; int stageFlip = 0;
    hipEvent_t sceneReady = nullptr;      // recorded behind the last in-place scene update; render launches wait for it on their stream
    bool havePrim = false; crt::PrimDev prim{}; uint32_t* dPrimTex = nullptr;       // crt_upload_primitive_scene: PrimitiveScene instead of a triangle scene
    int renderAccel = 0;                  // crt_set_render_accel: 0 = the scene's BVH / TLAS, CRT_ACCEL_KDTREE / CRT_ACCEL_GRID = Sample and Trace go through that structure
    crt::AltAccelDev alt{}; bool haveKd = false, haveGrid = false; std::vector<void*> altAllocs[2]; void* altTris = nullptr; uint32_t altTriCount = 0;   // KD-tree [0] / grid [1] buffers
    void* dQueryRays = nullptr; void* dQueryHits = nullptr; size_t queryCap = 0;      // crt_find_nearest staging (rays)
    // Latency mode of single-window launches (render_tiles_kernel's block table), driven by measurement — see next_block_table.  Stage 0 = the table solved from the cost probe's
    // estimates (one wavefront per tile when there was no probe); stages 1 .. kLatStages = tables solved from the tile costs the stage before measured; afterwards the fastest stage is used
    // (HIP event durations of the launches themselves; identical pixels whatever the table).
    static constexpr int kLatStages = 4;       // (round 3: the stages are solved, not stepped, and agree within launch-to-launch scatter from stage 1 on; round 2's stepping tuner needed 6)
    double tuneMs[kLatStages + 1] = {}; int tuneCount[kLatStages + 1] = {};
    hipEvent_t lastRenderEnd = nullptr;   // end event of the most recent render launch (owned by the timing lists)
    int latStage = 0;              // stage of the table on the device (0: none yet)
    int latBest = 0;               // fastest stage so far
    bool latDone = false;          // all stages measured, the fastest one's table is (being) installed
    bool latConfirming = false; std::vector<int> latQueue;      // after the last stage: the two fastest stages are timed once more
    bool latWarm = false;          // stage 0 has been measured once already (the first launch after an upload runs cold: it is measured twice)
    uint32_t latSlots = 0;         // wavefronts of render_tiles_kernel the device holds at once (0: not asked yet)
    bool latProbed = false;        // the cost probe has run for this camera / scene: stage 0 is a block table built from its estimates (latL[0]), not one wavefront per tile
    std::vector<uint8_t> latL[kLatStages + 1];          // lanes per wavefront of every tile, per stage ([0]: all 64)
    std::vector<uint32_t> latCost[kLatStages + 1];      // measured tile costs, per stage
    uint32_t* dTileCost = nullptr; uint32_t* hTileCost = nullptr; hipEvent_t costCopied = nullptr; bool costPending = false; int costStage = 0;
    uint32_t* dBlockDesc = nullptr; uint32_t* hBlockDesc = nullptr; uint32_t nBlocks = 0, nBlocksWide = 0, descCap = 0; hipEvent_t descReady = nullptr;   // table = [blocks of > 8 lanes: render_tiles_kernel][narrow blocks: render_narrow_kernel]
    // Jobs (launches of several windows): what each tile costs is measured once per camera / scene by the first job launch (every wavefront's duration, scaled to
    // 64 streams); later launches dispatch the tiles most expensive first and SPLIT: see split_point
    uint32_t* dJobCost = nullptr; uint32_t* hJobCost = nullptr; hipEvent_t jobCostCopied = nullptr; bool jobCostPending = false, jobCostValid = false;
    uint32_t recWaves = 0, recResident = 0, recWindows = 0; bool recPool = false;     // the measuring launch: wavefronts, how many the chip holds, windows, kernel
    double poolWindowTicks = 0;             // machine time of ONE window of this image under the pool, 100 MHz ticks (0: unknown)
    std::vector<uint32_t> jobCost;          // per local tile, 100 MHz ticks; sorted view = the tile order on the device once jobCostValid
    std::vector<uint32_t> jobOrder;
    uint32_t* dJobDesc = nullptr; uint32_t* hJobDesc = nullptr; uint32_t jobDescCap = 0, jobBlocks = 0, jobBlocksWide = 0, jobHead = 0; hipEvent_t jobDescReady = nullptr;
    uint32_t planWindows = 0, planFrames = 0; bool planPool = false, planValid = false;       // what the table on the device was planned for
    double jobTrialMs[2] = {0, 0};          // this plan's shape timed [0] plain (cost-ordered, no table) and [1] planned: the faster one is kept (planner_prepare)
    std::vector<hipEvent_t> splitEvents;    // end events of the second kernel of split launches (recycled round-robin)
    size_t splitSeq = 0; uint32_t splitLaunches = 0;
    uint64_t poolMinWaves = 65000; // launches of fewer (tile, 64-frame window) pairs run render_tiles_kernel: see crt_render
    bool usePool = true;          // render_pool_kernel (stream pool); CRT_RENDER_KERNEL=tiles selects render_tiles_kernel (one stream per lane)
    uint32_t ldsBytes = 0;
    // timing of the last crt_render
    std::vector<EventPair> evPool; size_t evUsedRender = 0, evUsedAcc = 0;
    std::deque<EventPair> evRender, evAcc;                        // launches not yet folded into the totals below (oldest first)
    double foldedRenderMs = 0, foldedAccMs = 0; uint32_t foldedLaunches = 0, poolLaunches = 0;

    int fail(int code, const char* fmt, ...)
    {
        char buf[1024];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
        err = buf; return code;
    }
    int hip(hipError_t e, const char* what)
    {
        if (e == hipSuccess) return 0;
        return fail(CRT_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e));
    }
    void freeAlt()
    {
        for (auto& v : altAllocs) { for (void* p : v) (void)hipFree(p); v.clear(); }
        if (altTris) (void)hipFree(altTris);
        altTris = nullptr; altTriCount = 0; haveKd = haveGrid = false; alt = crt::AltAccelDev{}; renderAccel = 0;
    }
    void freeScene()
    {
        for (void* p : sceneAllocs) (void)hipFree(p);
        sceneAllocs.clear(); haveScene = false;
        if (dPrimTex) { (void)hipFree(dPrimTex); dPrimTex = nullptr; }
        havePrim = false;
        freeAlt();                        // the alternative accelerators index the scene's triangles
    }
};

#define HIPCK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (ctx)->hip(e_, #call); } while (0)

namespace {

template <class T>
int upload(crt_ctx* c, const std::vector<T>& v, const T** out)
{
    *out = nullptr;
    if (v.empty()) return 0;
    void* d = nullptr;
    HIPCK(c, hipMalloc(&d, v.size() * sizeof(T)));
    c->sceneAllocs.push_back(d);
    HIPCK(c, hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));   // v is a temporary of the caller
    *out = reinterpret_cast<const T*>(d);
    return 0;
}

// height of the tree below node 0 in pushes: the ordered traversal pushes at most one sibling per interior level
int bvh_height(crt_ctx* c, const crt_bvh& b, uint32_t* heightOut)
{
    std::vector<std::pair<uint32_t, uint32_t>> st; st.push_back({0u, 0u});
    uint32_t h = 0; size_t visited = 0;
    while (!st.empty()) {
        auto [n, d] = st.back(); st.pop_back();
        if (++visited > (size_t)b.nodesUsed) return c->fail(CRT_ERR_INVALID, "BVH node graph is not a tree");
        const crt_bvh_node& nd = b.nodes[n];
        if (nd.triCount > 0) { if (d > h) h = d; continue; }
        if (nd.leftFirst == 0 || nd.leftFirst + 1 >= b.nodesUsed) return c->fail(CRT_ERR_INVALID, "BVH child index out of range (node %u)", n);
        st.push_back({nd.leftFirst, d + 1}); st.push_back({nd.leftFirst + 1, d + 1});
    }
    *heightOut = h;
    return 0;
}

// The treetop of a FileScene's BVH: its first `count` child pairs in breadth-first order from the root's pair, as a second copy whose child references name
// treetop entries (kRefTop | index) wherever the child's own pair is in the copy too.  Same boxes, same child order: a traversal that reads these records visits
// the reference's nodes in the reference's order.  `pairs` = the pair section, rootRef = packed reference of node 0 (interior).
void crt_build_treetop(const crt::NodePair* pairs, uint32_t nPairs, uint32_t rootRef, crt::NodePair* top, uint32_t count)
{
    std::vector<uint32_t> order; order.reserve(count);
    std::vector<uint32_t> slot(nPairs, 0xffffffffu);
    auto pair_of = [&](uint32_t ref) -> uint32_t { return ((ref & crt::kRefOffsetMask) << 4) / 64u; };     // interior reference -> index of its child pair
    order.push_back(pair_of(rootRef)); slot[order[0]] = 0;
    for (size_t i = 0; i < order.size() && order.size() < count; i++)
        for (int k = 0; k < 2 && order.size() < count; k++) {
            const uint32_t ref = pairs[order[i]].c[k].ref;
            if ((ref & 0xC0000000u) != crt::kRefInterior) continue;
            const uint32_t p = pair_of(ref);
            if (p < nPairs && slot[p] == 0xffffffffu) { slot[p] = (uint32_t)order.size(); order.push_back(p); }
        }
    for (uint32_t t = 0; t < count; t++) {
        if (t >= order.size()) { memset(&top[t], 0, sizeof(crt::NodePair)); continue; }
        top[t] = pairs[order[t]];
        for (int k = 0; k < 2; k++) {
            const uint32_t ref = top[t].c[k].ref;
            if ((ref & 0xC0000000u) == crt::kRefInterior) { const uint32_t p = pair_of(ref); if (p < nPairs && slot[p] != 0xffffffffu) top[t].c[k].ref = crt::kRefTop | slot[p]; }
        }
    }
}

} // namespace

extern "C" {

int crt_abi_version(void) { return CRT_ABI_VERSION; }

void crt_debug_enable_hooks(int on) { g_hooks = on != 0; }

int crt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* crt_last_error(crt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_createError.c_str(); }

int crt_create(crt_ctx** out, const crt_config* cfg)
{
    if (!out || !cfg) { g_createError = "crt_create: null argument"; return CRT_ERR_INVALID; }
    *out = nullptr;
    // independent launches overlap on several HIP streams; ROCm multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
    // (default 4) and serialises kernels that share one.  Only effective when the HIP runtime has not initialised yet
    // (a host that already uses HIP sets the variable itself); never overrides the user's value.
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    if (cfg->width < 16 || cfg->height < 16) { g_createError = "crt_create: width and height must be at least one 16x16 tile"; return CRT_ERR_INVALID; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        g_createError = "crt_create: no HIP device visible (this library has no CPU path)";
        return CRT_ERR_NO_DEVICE;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { g_createError = "crt_create: device ordinal out of range"; return CRT_ERR_INVALID; }
    crt_ctx* c = new crt_ctx();
    c->cfg = *cfg;
    if (const char* k = hook("CRT_RENDER_KERNEL")) {             // tests / A-B runs: "tiles" = never the stream pool, "pool_always" = also for launches of <= 64 frames
        c->usePool = strcmp(k, "tiles") != 0;
        if (!strcmp(k, "pool_always")) c->poolMinWaves = 0;
    }
    if (c->cfg.depthLimit < 0) c->cfg.depthLimit = 5;
    if (c->cfg.depthLimit > 5) { g_createError = "crt_create: depthLimit > 5 unsupported (throughput stack holds 5 factors; reference default is 5)"; delete c; return CRT_ERR_UNSUPPORTED; }
    if (c->cfg.maxFramesPerLaunch <= 0) c->cfg.maxFramesPerLaunch = c->cfg.collectStats ? 64 : 4096;   // 64 windows of 64 frames; per-tile clocks of a statistics context describe ONE window
    if (c->cfg.maxFramesPerLaunch > 64) c->cfg.maxFramesPerLaunch = c->cfg.maxFramesPerLaunch / 64 * 64;   // whole windows (< 64: one partial window per launch)
    if (c->cfg.renderStreams < 0) c->cfg.renderStreams = 0;
    c->tilesX = cfg->width / 16; c->tilesY = cfg->height / 16;      // truncating, as renderer.cpp:151
    const int tiles = c->tilesX * c->tilesY;
    int first = cfg->tileFirst, stride = cfg->tileStride <= 0 ? 1 : cfg->tileStride, count = cfg->tileCount;
    if (count < 0) { first = 0; stride = 1; count = tiles; }
    if (first < 0 || (count > 0 && (long long)first + (long long)(count - 1) * stride >= tiles)) {
        g_createError = "crt_create: tile range exceeds the image"; delete c; return CRT_ERR_INVALID;
    }
    c->tileFirst = (uint32_t)first; c->tileStride = (uint32_t)stride; c->tileCount = (uint32_t)count;

    auto bail = [&](hipError_t he, const char* what) {
        g_createError = std::string("crt_create: ") + what + ": " + hipGetErrorString(he);
        crt_destroy(c); return CRT_ERR_DEVICE;
    };
    if ((e = hipSetDevice(cfg->device)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    const size_t px = (size_t)cfg->width * cfg->height;
    if ((e = hipMalloc(&c->dAccOwned, px * 16)) != hipSuccess) return bail(e, "hipMalloc(accumulator)");
    c->dAcc = c->dAccOwned;
    if ((e = hipMemsetAsync(c->dAcc, 0, px * 16, c->stream)) != hipSuccess) return bail(e, "hipMemset(accumulator)");
    if ((e = hipMalloc((void**)&c->dPixels, px * 4)) != hipSuccess) return bail(e, "hipMalloc(pixels)");
    if ((e = hipMemsetAsync(c->dPixels, 0, px * 4, c->stream)) != hipSuccess) return bail(e, "hipMemset(pixels)");
    if ((e = hipMalloc((void**)&c->dTileSums, (size_t)tiles * 4)) != hipSuccess) return bail(e, "hipMalloc(tileSums)");
    if ((e = hipMemsetAsync(c->dTileSums, 0, (size_t)tiles * 4, c->stream)) != hipSuccess) return bail(e, "hipMemset(tileSums)");
    if ((e = hipMalloc((void**)&c->dCounters, sizeof(crt::Counters))) != hipSuccess) return bail(e, "hipMalloc(counters)");
    if ((e = hipMemsetAsync(c->dCounters, 0, sizeof(crt::Counters), c->stream)) != hipSuccess) return bail(e, "hipMemset(counters)");
    if ((e = hipMalloc((void**)&c->dQueryCursor, 64)) != hipSuccess) return bail(e, "hipMalloc(query cursor)");
    if (c->cfg.collectStats && count > 0) {
        if ((e = hipMalloc((void**)&c->dTileClocks, (size_t)count * 144)) != hipSuccess) return bail(e, "hipMalloc(tileClocks)");
        if ((e = hipMemsetAsync(c->dTileClocks, 0, (size_t)count * 144, c->stream)) != hipSuccess) return bail(e, "hipMemset(tileClocks)");
    }
    // Camera() defaults, template/camera.h:14-22
    memset(&c->hScene, 0, sizeof(c->hScene));
    const float aspect = (float)cfg->width / (float)cfg->height;
    crt::Scene& s = c->hScene;
    s.camPos[0] = 0; s.camPos[1] = 0; s.camPos[2] = -2;
    s.topLeft[0] = -aspect; s.topLeft[1] = 1; s.topLeft[2] = 0;
    s.topRight[0] = aspect; s.topRight[1] = 1; s.topRight[2] = 0;
    s.bottomLeft[0] = -aspect; s.bottomLeft[1] = -1; s.bottomLeft[2] = 0;
    s.W = cfg->width; s.H = cfg->height; s.invW = 1.0f / cfg->width; s.invH = 1.0f / cfg->height;
    s.depthLimit = c->cfg.depthLimit;
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    *out = c;
    return CRT_OK;
}

void crt_destroy(crt_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    for (auto st : c->streams) (void)hipStreamSynchronize(st);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->freeScene();
    for (auto& ev : c->evPool) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto& ev : c->evRender) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto& ev : c->evAcc) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    if (c->dAccOwned) (void)hipFree(c->dAccOwned);
    for (auto st : c->streams) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (c->narrowStream) { (void)hipStreamSynchronize(c->narrowStream); (void)hipStreamDestroy(c->narrowStream); }
    for (auto& r : c->inflight) (void)hipEventDestroy(r.freed);
    for (auto e : c->freeEvents) (void)hipEventDestroy(e);
    if (c->mustWait) (void)hipEventDestroy(c->mustWait);
    if (c->pool) (void)hipFree(c->pool);
    for (auto e : c->doneEvents) (void)hipEventDestroy(e);
    if (c->dPixels) (void)hipFree(c->dPixels);
    if (c->dTileSums) (void)hipFree(c->dTileSums);
    if (c->dCounters) (void)hipFree(c->dCounters);
    if (c->dQueryCursor) (void)hipFree(c->dQueryCursor);
    if (c->dTileClocks) (void)hipFree(c->dTileClocks);
    if (c->dTileOrder) (void)hipFree(c->dTileOrder);
    if (c->dTileCost) (void)hipFree(c->dTileCost);
    if (c->dJobCost) (void)hipFree(c->dJobCost);
    if (c->dJobDesc) (void)hipFree(c->dJobDesc);
    if (c->hJobDesc) (void)hipHostFree(c->hJobDesc);
    if (c->jobDescReady) (void)hipEventDestroy(c->jobDescReady);
    if (c->hJobCost) (void)hipHostFree(c->hJobCost);
    if (c->jobCostCopied) (void)hipEventDestroy(c->jobCostCopied);
    for (auto e : c->splitEvents) (void)hipEventDestroy(e);
    if (c->hTileCost) (void)hipHostFree(c->hTileCost);
    if (c->costCopied) (void)hipEventDestroy(c->costCopied);
    if (c->dBlockDesc) (void)hipFree(c->dBlockDesc);
    if (c->hBlockDesc) (void)hipHostFree(c->hBlockDesc);
    if (c->descReady) (void)hipEventDestroy(c->descReady);
    if (c->dQueryRays) (void)hipFree(c->dQueryRays);
    if (c->dQueryHits) (void)hipFree(c->dQueryHits);
    for (int k = 0; k < 2; k++) { if (c->hTileOrder[k]) (void)hipHostFree(c->hTileOrder[k]); if (c->orderCopied[k]) (void)hipEventDestroy(c->orderCopied[k]); }
    for (int k = 0; k < 2; k++) { if (c->hStage[k]) (void)hipHostFree(c->hStage[k]); if (c->stageCopied[k]) (void)hipEventDestroy(c->stageCopied[k]); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int crt_upload_scene(crt_ctx* c, const crt_scene_desc* sd)
{
    if (!c || !sd) return CRT_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->cfg.device));
    if (sd->kind != CRT_SCENE_FILE && sd->kind != CRT_SCENE_TLAS) return c->fail(CRT_ERR_INVALID, "unknown scene kind %d", sd->kind);
    if (!sd->bvhs || sd->bvhCount == 0) return c->fail(CRT_ERR_INVALID, "scene has no acceleration structure");
    if (sd->kind == CRT_SCENE_FILE && sd->bvhCount != 1) return c->fail(CRT_ERR_INVALID, "CRT_SCENE_FILE takes exactly one BVH (FileScene::acc)");
    if (sd->kind == CRT_SCENE_TLAS && (sd->bvhCount > 256 || !sd->tlasNodes || sd->tlasNodeCount < 2 * sd->bvhCount))
        return c->fail(CRT_ERR_INVALID, "TLAS scene needs <= 256 BLAS (tlas_bvh.cpp:21) and 2*blasCount TLAS nodes");
    if (sd->textureCount == 0 || !sd->textures) return c->fail(CRT_ERR_INVALID, "floor and skydome textures are required");
    if (sd->floorTexture < 0 || sd->floorTexture >= (int)sd->textureCount || sd->skyTexture < 0 || sd->skyTexture >= (int)sd->textureCount)
        return c->fail(CRT_ERR_INVALID, "floor/sky texture index out of range");
    for (uint32_t i = 0; i < sd->textureCount; i++)
        if (!sd->textures[i].pixels || sd->textures[i].width <= 0 || sd->textures[i].height <= 0) return c->fail(CRT_ERR_INVALID, "texture %u is empty", i);
    if (sd->materialCount > 0 && !sd->materials) return c->fail(CRT_ERR_INVALID, "materialCount is %u but materials is NULL", sd->materialCount);
    for (uint32_t i = 0; i < sd->materialCount; i++)
        if (sd->materials[i].texture >= (int)sd->textureCount) return c->fail(CRT_ERR_INVALID, "material %u: texture index out of range", i);
    if (sd->kind == CRT_SCENE_FILE) {
        if (!sd->objMatIdx || sd->objCount == 0) return c->fail(CRT_ERR_INVALID, "CRT_SCENE_FILE needs objMatIdx");
        for (uint32_t i = 0; i < sd->objCount; i++)
            if (sd->objMatIdx[i] < 0 || sd->objMatIdx[i] >= (int)sd->materialCount) return c->fail(CRT_ERR_INVALID, "objMatIdx entry out of range");
    }

    // ---- sizes of the sections of the geometry buffer: pairs | leaf tris (+1 pad record) | TLAS nodes | instances | shade records ----
    uint64_t nPairs = 0, nTris = 0;
    uint32_t maxHeight = 0;
    for (uint32_t bi = 0; bi < sd->bvhCount; bi++) {
        const crt_bvh& b = sd->bvhs[bi];
        if (!b.nodes || !b.triangles || !b.triangleIndices || b.nodesUsed == 0 || b.triCount == 0) return c->fail(CRT_ERR_INVALID, "BVH %u is empty", bi);
        if ((b.nodesUsed & 1u) == 0) return c->fail(CRT_ERR_INVALID, "BVH %u: nodesUsed must be odd (root + child pairs)", bi);
        if (sd->kind == CRT_SCENE_TLAS && (b.matIdx < 0 || b.matIdx >= (int)sd->materialCount)) return c->fail(CRT_ERR_INVALID, "BLAS %u: matIdx out of range", bi);
        if (sd->kind == CRT_SCENE_TLAS && b.objIdx != (int)bi + 2)
            return c->fail(CRT_ERR_INVALID, "BLAS %u: objIdx must be %u (TLASFileScene numbers objects from 2, tlas_file_scene.cpp:13,51-53)", bi, bi + 2);
        uint32_t h = 0; int r = bvh_height(c, b, &h); if (r) return r;
        if (h > maxHeight) maxHeight = h;
        nPairs += b.nodesUsed / 2; nTris += b.triCount;
    }
    const uint64_t pairBytes = nPairs ? nPairs * 64 : 64;                   // offset 0 must never be a leaf record (ref 0 = "done")
    // FileScene: a copy of the first pairs of the tree in breadth-first order (the "treetop") behind the pairs — render_narrow_kernel keeps it in LDS
    const uint32_t topCount = (sd->kind == CRT_SCENE_FILE && sd->bvhs[0].nodesUsed >= 3u) ? (uint32_t)std::min<uint64_t>(nPairs, crt::kTreetopMaxPairs) : 0u;
    const uint64_t topOffB = pairBytes;
    const uint64_t leafOffB = topOffB + (uint64_t)topCount * 64, leafBytes = (nTris + 1) * 48;       // +1: record fetches read 64 B from a 48-B LeafTri
    const uint64_t tlasOffB = (leafOffB + leafBytes + 63) & ~63ull;
    const uint64_t tlasBytes = (sd->kind == CRT_SCENE_TLAS) ? (uint64_t)sd->tlasNodeCount * 32 : 0;
    const uint64_t tlasPairOffB = (tlasOffB + tlasBytes + 63) & ~63ull;      // one NodePair-shaped record per TLAS interior node (at most tlasNodeCount / 2)
    const uint64_t tlasPairBytes = (sd->kind == CRT_SCENE_TLAS) ? (uint64_t)sd->tlasNodeCount * 32 : 0;
    const uint64_t instOffB = (tlasPairOffB + tlasPairBytes + 127) & ~127ull;
    const uint64_t instBytes = (sd->kind == CRT_SCENE_TLAS) ? (uint64_t)sd->bvhCount * 128 : 0;
    const uint64_t shadeOffB = (instOffB + instBytes + 63) & ~63ull;
    const uint64_t total = shadeOffB + nTris * 64;
    if (total >= crt::kMaxGeomBytes) return c->fail(CRT_ERR_UNSUPPORTED, "scene geometry needs %llu bytes; this build addresses 4 GiB", (unsigned long long)total);
    if (sd->kind == CRT_SCENE_TLAS && sd->tlasNodeCount > 0x7fffu) return c->fail(CRT_ERR_UNSUPPORTED, "TLAS node index exceeds 15 bits");

    std::vector<char> geom((size_t)total, 0);
    crt::NodePair* pairs = reinterpret_cast<crt::NodePair*>(geom.data());
    crt::LeafTri* leaf = reinterpret_cast<crt::LeafTri*>(geom.data() + leafOffB);
    crt::TlasNode* tlas = reinterpret_cast<crt::TlasNode*>(geom.data() + tlasOffB);
    crt::NodePair* tlasPairs = reinterpret_cast<crt::NodePair*>(geom.data() + tlasPairOffB);
    bool ref16ok = nPairs <= crt::kRef16MaxIndex && nTris < crt::kRef16MaxIndex;
    crt::Instance* inst = reinterpret_cast<crt::Instance*>(geom.data() + instOffB);
    crt::ShadeTri* shade = reinterpret_cast<crt::ShadeTri*>(geom.data() + shadeOffB);

    uint64_t pairBase = 0, triBase = 0; uint32_t rootRef0 = 0, rootRef0_16 = 0;
    for (uint32_t bi = 0; bi < sd->bvhCount; bi++) {
        const crt_bvh& b = sd->bvhs[bi];
        // packed reference of node n (layout.h): interior -> offset of its child pair, leaf -> offset of its first LeafTri, both in 16-byte units
        auto ref_of = [&](uint32_t n, uint32_t* ref) -> int {
            const crt_bvh_node& nd = b.nodes[n];
            if (nd.triCount > 0) {
                if ((uint64_t)nd.leftFirst + nd.triCount > b.triCount) return c->fail(CRT_ERR_INVALID, "leaf range out of bounds (BVH %u node %u)", bi, n);
                *ref = (uint32_t)((leafOffB + (triBase + nd.leftFirst) * 48) >> 4);
            } else {
                if ((nd.leftFirst & 1u) == 0) return c->fail(CRT_ERR_INVALID, "interior node %u: children must be allocated pairwise starting at an odd index", n);
                *ref = crt::kRefInterior | (uint32_t)(((pairBase + ((nd.leftFirst - 1u) >> 1)) * 64) >> 4);
            }
            return 0;
        };
        // the same reference in 16 bits (layout.h): record INDEX instead of offset; only meaningful when ref16ok
        auto ref16_of = [&](uint32_t n) -> uint32_t {
            const crt_bvh_node& nd = b.nodes[n];
            if (nd.triCount > 0) return (uint32_t)((triBase + nd.leftFirst + 1) & crt::kRef16IndexMask);
            return crt::kRef16Interior | (uint32_t)((pairBase + ((nd.leftFirst - 1u) >> 1)) & crt::kRef16IndexMask);
        };
        int r;
        for (uint32_t n = 1; n + 1 < b.nodesUsed; n += 2) {
            crt::NodePair& p = pairs[pairBase + ((n - 1) >> 1)];
            for (int k = 0; k < 2; k++) {
                const crt_bvh_node& nd = b.nodes[n + k];
                memcpy(p.c[k].lo, nd.aabbMin, 12); memcpy(p.c[k].hi, nd.aabbMax, 12);
                if ((r = ref_of(n + k, &p.c[k].ref))) return r;
                p.c[k].ref16 = ref16_of(n + k);
            }
        }
        uint32_t rootRef = 0; if ((r = ref_of(0, &rootRef))) return r;
        const uint32_t rootRef16 = ref16_of(0);
        // `remain` of every leaf slot: triangles left in its leaf including itself
        std::vector<uint32_t> remain(b.triCount, 1u);
        for (uint32_t n = 0; n < b.nodesUsed; n++) {
            const crt_bvh_node& nd = b.nodes[n];
            if (nd.triCount == 0) continue;
            for (uint32_t i = 0; i < nd.triCount; i++) remain[nd.leftFirst + i] = nd.triCount - i;
        }
        for (uint32_t j = 0; j < b.triCount; j++) {                       // leaf order: triangleIndices resolved here
            const uint32_t ti = b.triangleIndices[j];
            if (ti >= b.triCount) return c->fail(CRT_ERR_INVALID, "BVH %u: triangleIndices[%u] out of range", bi, j);
            const crt_tri& t = b.triangles[ti];
            crt::LeafTri& lt = leaf[triBase + j];
            for (int k = 0; k < 3; k++) { lt.v0[k] = t.vertex0[k]; lt.e1[k] = t.vertex1[k] - t.vertex0[k]; lt.e2[k] = t.vertex2[k] - t.vertex0[k]; }
            lt.shadeIdx = (uint32_t)(triBase + ti);
            lt.objIdx = (sd->kind == CRT_SCENE_TLAS) ? b.objIdx : t.objIdx;
            lt.remain = remain[j];
            if (sd->kind == CRT_SCENE_FILE && (t.objIdx < 2 || (uint32_t)(t.objIdx - 2) >= sd->objCount))
                return c->fail(CRT_ERR_INVALID, "triangle %u: objIdx %d has no entry in objMatIdx", ti, t.objIdx);
        }
        for (uint32_t ti = 0; ti < b.triCount; ti++) {                    // shading records in the reference's triIdx order
            const crt_tri& t = b.triangles[ti];
            if (sd->kind == CRT_SCENE_FILE && (t.objIdx < 2 || (uint32_t)(t.objIdx - 2) >= sd->objCount))   // every triangle, not only those triangleIndices reaches
                return c->fail(CRT_ERR_INVALID, "triangle %u: objIdx %d has no entry in objMatIdx", ti, t.objIdx);
            crt::ShadeTri& s = shade[triBase + ti];
            memcpy(s.n0, t.normal0, 12); memcpy(s.n1, t.normal1, 12); memcpy(s.n2, t.normal2, 12);
            memcpy(s.uv0, t.uv0, 8); memcpy(s.uv1, t.uv1, 8); memcpy(s.uv2, t.uv2, 8);
            // materials[models[tri.objIdx - 2]->matIdx] (file_scene.cpp:207) / materials[blas->matIdx] (tlas_file_scene.cpp:240); +2: [0] light, [1] floor
            s.mat = 2 + ((sd->kind == CRT_SCENE_TLAS) ? b.matIdx : sd->objMatIdx[t.objIdx - 2]);
        }
        if (sd->kind == CRT_SCENE_TLAS) {
            crt::Instance& in = inst[bi];
            memcpy(in.invT, b.invT, 48); memcpy(in.T, b.T, 48);
            in.shadeBase = (uint32_t)triBase; in.rootRef16 = rootRef16; in.rootRef = rootRef; in.objIdx = b.objIdx;
        } else { rootRef0 = rootRef; rootRef0_16 = rootRef16; }
        pairBase += b.nodesUsed / 2; triBase += b.triCount;
    }
    leaf[nTris].remain = 1;                                                 // pad record
    if (topCount) crt_build_treetop(pairs, (uint32_t)nPairs, rootRef0, reinterpret_cast<crt::NodePair*>(geom.data() + topOffB), topCount);
    uint32_t tlasHeight = 0, tlasRoot = 0, tlasRoot16 = 0;
    if (sd->kind == CRT_SCENE_TLAS) {
        uint32_t nTlasPairs = 0;
        for (uint32_t i = 0; i < sd->tlasNodeCount; i++) {      // device copy carries each node's packed reference instead of leftRight / BLAS
            const crt_tlas_node& nd = sd->tlasNodes[i];
            memcpy(tlas[i].lo, nd.aabbMin, 12); memcpy(tlas[i].hi, nd.aabbMax, 12);
            tlas[i].ref = nd.leftRight ? (crt::kRefTlasInterior | (nd.leftRight & 0x7fffu) | (((nd.leftRight >> 16) & 0x7fffu) << 15))
                                       : (crt::kRefTlasLeaf | (nd.BLAS & 0xffffu));
            tlas[i].ref16 = nd.leftRight ? (crt::kRef16TlasBit | nTlasPairs++) : (crt::kRef16TlasLeaf | (nd.BLAS & crt::kRef16IndexMask));   // interior nodes number their child pairs
        }
        std::vector<std::pair<uint32_t, uint32_t>> st; st.push_back({0u, 0u}); size_t visited = 0;
        while (!st.empty()) {
            auto [n, d] = st.back(); st.pop_back();
            if (++visited > (size_t)sd->tlasNodeCount) return c->fail(CRT_ERR_INVALID, "TLAS node graph is not a tree");
            const crt_tlas_node& nd = sd->tlasNodes[n];
            if (nd.leftRight == 0) { if (nd.BLAS >= sd->bvhCount) return c->fail(CRT_ERR_INVALID, "TLAS leaf references BLAS %u", nd.BLAS); if (d > tlasHeight) tlasHeight = d; continue; }
            const uint32_t l = nd.leftRight & 0xffffu, r = nd.leftRight >> 16;
            if (l >= sd->tlasNodeCount || r >= sd->tlasNodeCount) return c->fail(CRT_ERR_INVALID, "TLAS child index out of range");
            st.push_back({l, d + 1}); st.push_back({r, d + 1});
        }
        // the child pair of every TLAS interior node, side by side in the NodePair layout (render_pool_kernel fetches it with one 64-byte record load)
        for (uint32_t i = 0; i < sd->tlasNodeCount; i++) {
            const crt_tlas_node& nd = sd->tlasNodes[i];
            if (nd.leftRight == 0) continue;
            crt::NodePair& p = tlasPairs[tlas[i].ref16 & crt::kRef16IndexMask];
            memcpy(&p.c[0], &tlas[nd.leftRight & 0xffffu], 32); memcpy(&p.c[1], &tlas[nd.leftRight >> 16], 32);
        }
        tlasRoot = tlas[0].ref; tlasRoot16 = tlas[0].ref16;
    }
    // texel pool + materials ([0] light, [1] floor, then the scene's — file_scene.cpp:10-12, 30-38); each carries its texture descriptor
    std::vector<uint32_t> texOff(sd->textureCount); uint64_t texels = 0;
    for (uint32_t i = 0; i < sd->textureCount; i++) {
        texOff[i] = (uint32_t)texels;
        texels += (uint64_t)sd->textures[i].width * sd->textures[i].height;
        if (texels > 0xffffffffull) return c->fail(CRT_ERR_UNSUPPORTED, "texture pool exceeds 2^32 texels");
    }
    auto bindTex = [&](crt::Material& m, int t) {
        if (t < 0) { m.texOffset = 0; m.texW = 0; m.texH = 0; }
        else { m.texOffset = texOff[t]; m.texW = sd->textures[t].width; m.texH = sd->textures[t].height; }
    };
    std::vector<crt::Material> mats(2 + sd->materialCount);
    memset(mats.data(), 0, mats.size() * sizeof(crt::Material));
    bindTex(mats[0], -1);
    bindTex(mats[1], sd->floorTexture);
    for (uint32_t i = 0; i < sd->materialCount; i++) {
        crt::Material& m = mats[2 + i];
        m.reflectivity = sd->materials[i].reflectivity; m.refractivity = sd->materials[i].refractivity;
        memcpy(m.absorption, sd->materials[i].absorption, 12);
        bindTex(m, sd->materials[i].texture);
    }

    // launches still in flight read the previous scene's buffers (render streams first: the main stream's accumulates wait on them)
    for (auto st : c->streams) HIPCK(c, hipStreamSynchronize(st));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->freeScene();
    uint32_t* dTexels = nullptr;
    HIPCK(c, hipMalloc((void**)&dTexels, (size_t)texels * 4)); c->sceneAllocs.push_back(dTexels);
    for (uint32_t i = 0; i < sd->textureCount; i++)
        HIPCK(c, hipMemcpyAsync(dTexels + texOff[i], sd->textures[i].pixels, (size_t)sd->textures[i].width * sd->textures[i].height * 4, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));

    crt::Scene& s = c->hScene;
    s.kind = sd->kind;
    memcpy(s.lightInvT, sd->lightInvT, 48);
    s.lightNrm[0] = -sd->lightT[1]; s.lightNrm[1] = -sd->lightT[5]; s.lightNrm[2] = -sd->lightT[9];   // Quad::GetNormal, primitives.h:363-367
    s.lightSize = sd->lightSize;
    {   // GetLightPos (file_scene.cpp:156-162): middle of the quad's two corners, 0.01 below; scalar TransformPosition order
        const float* m = sd->lightT;
        const float ax = -0.5f, ay = 0.0f, az = -0.5f, bx = 0.5f, by = 0.0f, bz = 0.5f;
        const float c1[3] = {m[0] * ax + m[1] * ay + m[2] * az + m[3] * 1.0f, m[4] * ax + m[5] * ay + m[6] * az + m[7] * 1.0f, m[8] * ax + m[9] * ay + m[10] * az + m[11] * 1.0f};
        const float c2[3] = {m[0] * bx + m[1] * by + m[2] * bz + m[3] * 1.0f, m[4] * bx + m[5] * by + m[6] * bz + m[7] * 1.0f, m[8] * bx + m[9] * by + m[10] * bz + m[11] * 1.0f};
        s.lightPos[0] = (c1[0] + c2[0]) * 0.5f - 0.0f; s.lightPos[1] = (c1[1] + c2[1]) * 0.5f - 0.01f; s.lightPos[2] = (c1[2] + c2[2]) * 0.5f - 0.0f;
    }
    memcpy(s.floorN, sd->floorN, 12); s.floorD = sd->floorD; s.floorInvto = sd->floorInvto;
    {   // what FileScene / TLASFileScene always build: a translated, unrotated quad and the y-up floor plane
        const float* m = sd->lightInvT;
        s.lightAxis = (m[0] == 1.0f && m[1] == 0.0f && m[2] == 0.0f && m[4] == 0.0f && m[5] == 1.0f && m[6] == 0.0f && m[8] == 0.0f && m[9] == 0.0f && m[10] == 1.0f) ? 1u : 0u;
        s.floorAxisY = (sd->floorN[0] == 0.0f && sd->floorN[1] == 1.0f && sd->floorN[2] == 0.0f) ? 1u : 0u;
        if (hook("CRT_DEBUG_GENERAL_PRIMS")) s.lightAxis = s.floorAxisY = 0u;     // tests: the general quad / plane expressions on the standard scenes
    }
    s.floorMat = mats[1];
    s.skyOffset = texOff[sd->skyTexture]; s.skyW = sd->textures[sd->skyTexture].width; s.skyH = sd->textures[sd->skyTexture].height;
    s.texels = dTexels;
    int r;
    const char* dGeom = nullptr;
    if ((r = upload(c, geom, &dGeom))) return r;
    s.geom = dGeom;
    s.tlasOff = (uint32_t)tlasOffB; s.instOff = (uint32_t)instOffB; s.shadeOff = (uint32_t)shadeOffB;
    s.leafOff = (uint32_t)leafOffB; s.tlasPairOff = (uint32_t)tlasPairOffB;
    s.topOff = (uint32_t)topOffB; s.topCount = topCount;
    s.rootRef16 = (sd->kind == CRT_SCENE_TLAS) ? tlasRoot16 : rootRef0_16;
    s.ref16ok = (ref16ok && !hook("CRT_DEBUG_NO_REF16")) ? 1u : 0u;
    if ((r = upload(c, mats, &s.mats))) return r;
    s.rootRef = (sd->kind == CRT_SCENE_TLAS) ? tlasRoot : rootRef0;
    // Traversal stack entries (LDS is what limits the waves per SIMD, so no slack): the ordered traversal keeps at most one pending sibling per level
    // below the root, i.e. <= height entries, and its dead store (the far child is written to slot sp before the push is decided) happens at an
    // interior node, where sp <= height - 1.  Two-level scenes add the TLAS entries and the return marker below the BLAS part.
    s.bvhStack = maxHeight > 0 ? maxHeight : 1;
    // the root's two children also travel in the kernel arguments: the render kernel takes every ray's first traversal step from scalar registers
    s.rootIsPair = 0; memset(s.rootPair, 0, sizeof(s.rootPair));
    if (sd->kind == CRT_SCENE_TLAS) {
        if ((tlasRoot & 0xC0000000u) == crt::kRefTlasInterior) {
            memcpy(s.rootPair, &tlas[tlasRoot & 0x7fffu], 32); memcpy(s.rootPair + 8, &tlas[(tlasRoot >> 15) & 0x7fffu], 32);
            s.rootIsPair = 1;
        }
    } else if (rootRef0 & crt::kRefInterior) {
        memcpy(s.rootPair, geom.data() + ((size_t)(rootRef0 & crt::kRefOffsetMask) << 4), 64);
        s.rootIsPair = 1;
    }
    if (hook("CRT_DEBUG_NO_ROOTPAIR")) s.rootIsPair = 0;                            // tests: every ray starts at the root reference instead
    s.stackDepth = s.bvhStack + ((sd->kind == CRT_SCENE_TLAS) ? tlasHeight + 1 : 0);   // + TLAS pushes + the return marker
    c->ldsBytes = s.stackDepth * 64u * 4u; c->latSlots = 0;
    if (const char* e = hook("CRT_DEBUG_EXTRA_LDS")) c->ldsBytes += (uint32_t)atoi(e);   // occupancy experiments only
    if (c->ldsBytes + 15u * 256u > 64u * 1024u) return c->fail(CRT_ERR_UNSUPPORTED, "tree height %u (+TLAS %u) needs %u bytes of LDS traversal stack per wave (> 64 KiB)", maxHeight, tlasHeight, c->ldsBytes);
    // world-space bounds of all meshes (FileScene: root box of the BVH; TLAS: root box of the TLAS) for the dispatch-order heuristic
    {
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        auto grow = [&](const float* mn, const float* mx) { for (int k = 0; k < 3; k++) { if (mn[k] < lo[k]) lo[k] = mn[k]; if (mx[k] > hi[k]) hi[k] = mx[k]; } };
        if (sd->kind == CRT_SCENE_TLAS) grow(sd->tlasNodes[0].aabbMin, sd->tlasNodes[0].aabbMax);
        else grow(sd->bvhs[0].nodes[0].aabbMin, sd->bvhs[0].nodes[0].aabbMax);
        memcpy(c->meshLo, lo, 12); memcpy(c->meshHi, hi, 12); c->orderDirty = true;
    }
    {   // keep what crt_update_scene needs
        crt_ctx::Flat& f = c->flat;
        f.topOff = topOffB; f.topCount = topCount; f.rootRef = rootRef0;
        f.kind = sd->kind; f.leafOff = leafOffB; f.tlasOff = tlasOffB; f.tlasPairOff = tlasPairOffB; f.instOff = instOffB; f.shadeOff = shadeOffB;
        f.tlasNodeCount = (sd->kind == CRT_SCENE_TLAS) ? sd->tlasNodeCount : 0; f.maxHeight = maxHeight;
        f.pairBase.clear(); f.triBase.clear(); f.nodesUsed.clear(); f.triCount.clear();
        uint64_t pb = 0, tb = 0;
        for (uint32_t bi = 0; bi < sd->bvhCount; bi++) { f.pairBase.push_back(pb); f.triBase.push_back(tb); f.nodesUsed.push_back(sd->bvhs[bi].nodesUsed); f.triCount.push_back(sd->bvhs[bi].triCount); pb += sd->bvhs[bi].nodesUsed / 2; tb += sd->bvhs[bi].triCount; }
        f.geom.swap(geom);
    }
    c->haveScene = true;
    return CRT_OK;
}

// fills the TLAS sections of a host geometry image from the reference's TLASBVHNode array: per-node records with both reference forms, and the child
// pair of every interior node side by side (NodePair layout).  Returns the TLAS height, or a negative status.
static int flatten_tlas(crt_ctx* c, const crt_tlas_node* nodes, uint32_t count, uint32_t bvhCount, crt::TlasNode* tlas, crt::NodePair* tlasPairs, uint32_t* heightOut)
{
    uint32_t nTlasPairs = 0;
    for (uint32_t i = 0; i < count; i++) {      // device copy carries each node's packed reference instead of leftRight / BLAS
        const crt_tlas_node& nd = nodes[i];
        memcpy(tlas[i].lo, nd.aabbMin, 12); memcpy(tlas[i].hi, nd.aabbMax, 12);
        tlas[i].ref = nd.leftRight ? (crt::kRefTlasInterior | (nd.leftRight & 0x7fffu) | (((nd.leftRight >> 16) & 0x7fffu) << 15))
                                   : (crt::kRefTlasLeaf | (nd.BLAS & 0xffffu));
        tlas[i].ref16 = nd.leftRight ? (crt::kRef16TlasBit | nTlasPairs++) : (crt::kRef16TlasLeaf | (nd.BLAS & crt::kRef16IndexMask));   // interior nodes number their child pairs
    }
    std::vector<std::pair<uint32_t, uint32_t>> st; st.push_back({0u, 0u}); size_t visited = 0; uint32_t height = 0;
    while (!st.empty()) {
        auto [n, d] = st.back(); st.pop_back();
        if (++visited > (size_t)count) return c->fail(CRT_ERR_INVALID, "TLAS node graph is not a tree");
        const crt_tlas_node& nd = nodes[n];
        if (nd.leftRight == 0) { if (nd.BLAS >= bvhCount) return c->fail(CRT_ERR_INVALID, "TLAS leaf references BLAS %u", nd.BLAS); if (d > height) height = d; continue; }
        const uint32_t l = nd.leftRight & 0xffffu, r = nd.leftRight >> 16;
        if (l >= count || r >= count) return c->fail(CRT_ERR_INVALID, "TLAS child index out of range");
        st.push_back({l, d + 1}); st.push_back({r, d + 1});
    }
    for (uint32_t i = 0; i < count; i++) {
        const crt_tlas_node& nd = nodes[i];
        if (nd.leftRight == 0) continue;
        crt::NodePair& p = tlasPairs[tlas[i].ref16 & crt::kRef16IndexMask];
        memcpy(&p.c[0], &tlas[nd.leftRight & 0xffffu], 32); memcpy(&p.c[1], &tlas[nd.leftRight >> 16], 32);
    }
    *heightOut = height;
    return 0;
}

// the Scene block's view of the root: its child pair travels in the kernel arguments (see render kernels), plus the dispatch-order bounds
static void set_root(crt_ctx* c, int kind, const float* lo, const float* hi)
{
    crt::Scene& s = c->hScene; const crt_ctx::Flat& f = c->flat;
    s.rootIsPair = 0; memset(s.rootPair, 0, sizeof(s.rootPair));
    if (kind == CRT_SCENE_TLAS) {
        const crt::TlasNode* tlas = reinterpret_cast<const crt::TlasNode*>(f.geom.data() + f.tlasOff);
        const uint32_t root = tlas[0].ref;
        s.rootRef = root; s.rootRef16 = tlas[0].ref16;
        if ((root & 0xC0000000u) == crt::kRefTlasInterior) { memcpy(s.rootPair, &tlas[root & 0x7fffu], 32); memcpy(s.rootPair + 8, &tlas[(root >> 15) & 0x7fffu], 32); s.rootIsPair = 1; }
    } else if (s.rootRef & crt::kRefInterior) {
        memcpy(s.rootPair, f.geom.data() + ((size_t)(s.rootRef & crt::kRefOffsetMask) << 4), 64);
        s.rootIsPair = 1;
    }
    if (hook("CRT_DEBUG_NO_ROOTPAIR")) s.rootIsPair = 0;
    memcpy(c->meshLo, lo, 12); memcpy(c->meshHi, hi, 12); c->orderDirty = true;
}

int crt_update_scene(crt_ctx* c, const crt_scene_desc* sd, uint32_t what)
{
    if (!c || !sd) return CRT_ERR_INVALID;
    if (!c->haveScene) return c->fail(CRT_ERR_STATE, "crt_update_scene before crt_upload_scene");
    crt_ctx::Flat& f = c->flat;
    if (sd->kind != f.kind || sd->bvhCount != f.nodesUsed.size() || !sd->bvhs) return c->fail(CRT_ERR_INVALID, "crt_update_scene: scene kind / BVH count differ from the uploaded scene");
    if ((what & ~(uint32_t)(CRT_UPDATE_TRANSFORMS | CRT_UPDATE_BOUNDS)) || what == 0) return c->fail(CRT_ERR_INVALID, "crt_update_scene: unknown update flags");
    if ((what & CRT_UPDATE_TRANSFORMS) && f.kind != CRT_SCENE_TLAS) return c->fail(CRT_ERR_INVALID, "CRT_UPDATE_TRANSFORMS applies to two-level scenes (a FileScene bakes its transforms into the triangles)");
    HIPCK(c, hipSetDevice(c->cfg.device));
    for (uint32_t bi = 0; bi < sd->bvhCount; bi++)
        if (sd->bvhs[bi].nodesUsed != f.nodesUsed[bi] || sd->bvhs[bi].triCount != f.triCount[bi] || !sd->bvhs[bi].nodes || !sd->bvhs[bi].triangles || !sd->bvhs[bi].triangleIndices)
            return c->fail(CRT_ERR_INVALID, "crt_update_scene: BVH %u changed its topology (node / triangle count); upload the scene again", bi);
    // ---- every check first: a refused update leaves the host mirror (and so the next successful update) untouched ----
    if (f.kind == CRT_SCENE_TLAS && (!sd->tlasNodes || sd->tlasNodeCount != f.tlasNodeCount))
        return c->fail(CRT_ERR_INVALID, "crt_update_scene: a two-level scene needs the node array of the TLASBVH::Build that followed the change (tlasNodes, %u nodes as uploaded)", f.tlasNodeCount);
    if (what & CRT_UPDATE_BOUNDS) {
        const crt::LeafTri* leaf = reinterpret_cast<const crt::LeafTri*>(f.geom.data() + f.leafOff);
        for (uint32_t bi = 0; bi < sd->bvhCount; bi++) {
            const crt_bvh& b = sd->bvhs[bi];
            for (uint32_t j = 0; j < b.triCount; j++) {
                const uint32_t ti = b.triangleIndices[j];
                if (ti >= b.triCount) return c->fail(CRT_ERR_INVALID, "BVH %u: triangleIndices[%u] out of range", bi, j);
                if (leaf[f.triBase[bi] + j].shadeIdx != (uint32_t)(f.triBase[bi] + ti)) return c->fail(CRT_ERR_INVALID, "BVH %u: triangleIndices changed; Refit keeps the leaf order — upload the scene again", bi);
            }
        }
    }
    uint32_t tlasHeight = 0;
    std::vector<char> tlasImage;                                            // the TLAS sections (nodes + child pairs), flattened aside and copied in once they are known to be valid
    if (f.kind == CRT_SCENE_TLAS) {
        tlasImage.assign((size_t)(f.instOff - f.tlasOff), 0);
        int r = flatten_tlas(c, sd->tlasNodes, sd->tlasNodeCount, sd->bvhCount, reinterpret_cast<crt::TlasNode*>(tlasImage.data()),
                             reinterpret_cast<crt::NodePair*>(tlasImage.data() + (f.tlasPairOff - f.tlasOff)), &tlasHeight);
        if (r) return r;
        if ((c->hScene.bvhStack + tlasHeight + 1) * 64u * 4u + 15u * 256u > 64u * 1024u)
            return c->fail(CRT_ERR_UNSUPPORTED, "TLAS height %u needs %u bytes of LDS traversal stack per wave (> 64 KiB)", tlasHeight, (c->hScene.bvhStack + tlasHeight + 1) * 256u);
    }
    // ---- apply ----
    size_t lo = SIZE_MAX, hi = 0;                                         // byte range of the geometry buffer to rewrite
    auto touch = [&](size_t a, size_t b) { if (a < lo) lo = a; if (b > hi) hi = b; };
    if (what & CRT_UPDATE_BOUNDS) {
        // BVH::Refit / BLASBVH::Refit (bvh.cpp:26-43): same tree, new boxes and vertex positions.  References, leaf order and shading records stay.
        crt::NodePair* pairs = reinterpret_cast<crt::NodePair*>(f.geom.data());
        crt::LeafTri* leaf = reinterpret_cast<crt::LeafTri*>(f.geom.data() + f.leafOff);
        for (uint32_t bi = 0; bi < sd->bvhCount; bi++) {
            const crt_bvh& b = sd->bvhs[bi];
            for (uint32_t n = 1; n + 1 < b.nodesUsed; n += 2) {
                crt::NodePair& p = pairs[f.pairBase[bi] + ((n - 1) >> 1)];
                for (int k = 0; k < 2; k++) { memcpy(p.c[k].lo, b.nodes[n + k].aabbMin, 12); memcpy(p.c[k].hi, b.nodes[n + k].aabbMax, 12); }
            }
            for (uint32_t j = 0; j < b.triCount; j++) {
                const crt_tri& t = b.triangles[b.triangleIndices[j]];
                crt::LeafTri& lt = leaf[f.triBase[bi] + j];
                for (int k = 0; k < 3; k++) { lt.v0[k] = t.vertex0[k]; lt.e1[k] = t.vertex1[k] - t.vertex0[k]; lt.e2[k] = t.vertex2[k] - t.vertex0[k]; }
            }
        }
        if (f.topCount) crt_build_treetop(pairs, (uint32_t)(f.topOff / 64), f.rootRef, reinterpret_cast<crt::NodePair*>(f.geom.data() + f.topOff), f.topCount);   // same membership, new boxes
        touch(0, (size_t)f.tlasOff);
    }
    if (f.kind == CRT_SCENE_TLAS) {
        if (what & CRT_UPDATE_TRANSFORMS) {
            crt::Instance* inst = reinterpret_cast<crt::Instance*>(f.geom.data() + f.instOff);
            for (uint32_t bi = 0; bi < sd->bvhCount; bi++) { memcpy(inst[bi].invT, sd->bvhs[bi].invT, 48); memcpy(inst[bi].T, sd->bvhs[bi].T, 48); }   // BLASBVH::SetTransform, blas_bvh.cpp:363-374
        }
        // TLASBVH::Build (tlas_bvh.cpp:17-55) ran on the host after SetTransform / Refit: new node array of the same size
        memcpy(f.geom.data() + f.tlasOff, tlasImage.data(), tlasImage.size());
        touch((size_t)f.tlasOff, (size_t)f.shadeOff);
        crt::Scene& s = c->hScene;
        s.stackDepth = s.bvhStack + tlasHeight + 1;
        c->ldsBytes = s.stackDepth * 64u * 4u; c->latSlots = 0;
    }
    if (lo >= hi) return CRT_OK;
    // In-place rewrite, no allocation of device memory and no host wait for the GPU: the copy runs on the main stream, which is ordered behind every
    // render launch submitted so far (it waits for each launch's end event before that launch's accumulate); launches submitted later wait
    // for `sceneReady` on their own stream.  Pinned staging buffers alternate and grow on demand; one is reused only after its own copy.
    const int k = c->stageFlip ^= 1;
    const size_t bytes = hi - lo;
    if (c->stageBytes[k] < bytes) {
        if (c->hStage[k]) { HIPCK(c, hipEventSynchronize(c->stageCopied[k])); HIPCK(c, hipHostFree(c->hStage[k])); c->hStage[k] = nullptr; }
        HIPCK(c, hipHostMalloc((void**)&c->hStage[k], bytes, hipHostMallocDefault)); c->stageBytes[k] = bytes;
        if (!c->stageCopied[k]) HIPCK(c, hipEventCreateWithFlags(&c->stageCopied[k], hipEventDisableTiming));
    } else HIPCK(c, hipEventSynchronize(c->stageCopied[k]));
    memcpy(c->hStage[k], f.geom.data() + lo, bytes);
    HIPCK(c, hipMemcpyAsync(const_cast<char*>(c->hScene.geom) + lo, c->hStage[k], bytes, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipEventRecord(c->stageCopied[k], c->stream));
    c->sceneReady = c->stageCopied[k];
    const float* blo = (f.kind == CRT_SCENE_TLAS) ? sd->tlasNodes[0].aabbMin : sd->bvhs[0].nodes[0].aabbMin;
    const float* bhi = (f.kind == CRT_SCENE_TLAS) ? sd->tlasNodes[0].aabbMax : sd->bvhs[0].nodes[0].aabbMax;
    set_root(c, f.kind, blo, bhi);
    return CRT_OK;
}

int crt_set_camera(crt_ctx* c, const float camPos[3], const float tl[3], const float tr[3], const float bl[3])
{
    if (!c || !camPos || !tl || !tr || !bl) return CRT_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->cfg.device));
    crt::Scene& s = c->hScene;
    if (!memcmp(s.camPos, camPos, 12) && !memcmp(s.topLeft, tl, 12) && !memcmp(s.topRight, tr, 12) && !memcmp(s.bottomLeft, bl, 12)) return CRT_OK;   // unchanged (a per-frame PushCamera)
    memcpy(s.camPos, camPos, 12); memcpy(s.topLeft, tl, 12); memcpy(s.topRight, tr, 12); memcpy(s.bottomLeft, bl, 12);
    c->orderDirty = true;
    return CRT_OK;      // the Scene block travels by value in every launch's kernel arguments
}

// the dispatch order of the tiles (local indices) to the device.  No host synchronisation: the copy runs on the main stream, which is ordered behind every render
// launch submitted so far (it waits for each launch's end event before that launch's accumulate), so the previous order is no longer read when it is overwritten;
// later launches wait for `orderReady` on their own stream.  The staging buffers are pinned and alternate; one is reused only after its own copy.
static int upload_tile_order(crt_ctx* c, const std::vector<uint32_t>& order)
{
    if (!c->dTileOrder) HIPCK(c, hipMalloc((void**)&c->dTileOrder, (size_t)c->tileCount * 4));
    const int k = c->orderFlip ^= 1;
    if (!c->hTileOrder[k]) {
        HIPCK(c, hipHostMalloc((void**)&c->hTileOrder[k], (size_t)c->tileCount * 4, hipHostMallocDefault));
        HIPCK(c, hipEventCreateWithFlags(&c->orderCopied[k], hipEventDisableTiming));
    } else HIPCK(c, hipEventSynchronize(c->orderCopied[k]));
    memcpy(c->hTileOrder[k], order.data(), order.size() * 4);
    HIPCK(c, hipMemcpyAsync(c->dTileOrder, c->hTileOrder[k], order.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipEventRecord(c->orderCopied[k], c->stream));
    c->orderReady = c->orderCopied[k];
    return 0;
}

// Tiles (local indices 0..tileCount) ordered so that those inside the screen-space bounding rectangle of the meshes' world box come first
// (the stream-pool kernel and wide launches dispatch in this order).  Pure scheduling heuristic: the projection uses the pin-hole camera of crt_set_camera in double precision and is conservative on failure
// (a corner behind the eye makes every tile a candidate).
static int update_tile_order(crt_ctx* c)
{
    if (!c->orderDirty || c->tileCount == 0) return 0;
    const crt::Scene& s = c->hScene;
    double R[3], Dn[3], N[3];
    for (int k = 0; k < 3; k++) { R[k] = (double)s.topRight[k] - s.topLeft[k]; Dn[k] = (double)s.bottomLeft[k] - s.topLeft[k]; }
    N[0] = R[1] * Dn[2] - R[2] * Dn[1]; N[1] = R[2] * Dn[0] - R[0] * Dn[2]; N[2] = R[0] * Dn[1] - R[1] * Dn[0];
    const double rr = R[0] * R[0] + R[1] * R[1] + R[2] * R[2], dd = Dn[0] * Dn[0] + Dn[1] * Dn[1] + Dn[2] * Dn[2];
    double E[3]; for (int k = 0; k < 3; k++) E[k] = (double)s.topLeft[k] - s.camPos[k];
    const double num = E[0] * N[0] + E[1] * N[1] + E[2] * N[2];
    // screen rectangle (in tiles, one tile of margin) of 8 world-space corners; false when a corner is behind the eye
    auto rect_of = [&](const float* P8, int r[4]) -> bool {
        double u0 = 1e30, u1 = -1e30, v0 = 1e30, v1 = -1e30;
        for (int i = 0; i < 8; i++) {
            const float* P = P8 + 3 * i;
            double d[3] = {P[0] - s.camPos[0], P[1] - s.camPos[1], P[2] - s.camPos[2]};
            const double den = d[0] * N[0] + d[1] * N[1] + d[2] * N[2];
            const double t = den != 0 ? num / den : -1;
            if (!(t > 0) || rr == 0 || dd == 0) return false;
            double Q[3]; for (int k = 0; k < 3; k++) Q[k] = s.camPos[k] + t * d[k] - s.topLeft[k];
            const double u = (Q[0] * R[0] + Q[1] * R[1] + Q[2] * R[2]) / rr, v = (Q[0] * Dn[0] + Q[1] * Dn[1] + Q[2] * Dn[2]) / dd;
            if (u < u0) u0 = u; if (u > u1) u1 = u; if (v < v0) v0 = v; if (v > v1) v1 = v;
        }
        r[0] = (int)floor(u0 * c->cfg.width / 16.0) - 1; r[1] = (int)floor(u1 * c->cfg.width / 16.0) + 1;
        r[2] = (int)floor(v0 * c->cfg.height / 16.0) - 1; r[3] = (int)floor(v1 * c->cfg.height / 16.0) + 1;
        return true;
    };
    float world[24];
    for (int i = 0; i < 8; i++) { world[3 * i] = (i & 1) ? c->meshHi[0] : c->meshLo[0]; world[3 * i + 1] = (i & 2) ? c->meshHi[1] : c->meshLo[1]; world[3 * i + 2] = (i & 4) ? c->meshHi[2] : c->meshLo[2]; }
    int wr[4] = {0, c->tilesX - 1, 0, c->tilesY - 1};
    (void)rect_of(world, wr);                                     // on failure wr stays the whole image
    std::vector<uint32_t> first, second, rest;
    for (uint32_t i = 0; i < c->tileCount; i++) {
        const uint32_t tile = c->tileFirst + i * c->tileStride;
        const int tx = (int)(tile % (uint32_t)c->tilesX), ty = (int)(tile / (uint32_t)c->tilesX);
        const bool inRect = tx >= wr[0] && tx <= wr[1] && ty >= wr[2] && ty <= wr[3];
        if (inRect) first.push_back(i); else rest.push_back(i);
    }
    // a new camera / scene: the latency mode measures again (see next_block_table)
    for (int k = 0; k <= crt_ctx::kLatStages; k++) { c->tuneCount[k] = 0; c->tuneMs[k] = 0; }
    c->latStage = 0; c->latBest = 0; c->latDone = false; c->latWarm = false; c->latConfirming = false; c->latQueue.clear(); c->costPending = false; c->latProbed = false;
    first.insert(first.end(), rest.begin(), rest.end());
    c->jobCostValid = false; c->jobCostPending = false; c->planValid = false;
    { const int r = upload_tile_order(c, first); if (r) return r; }
    c->orderDirty = false;
    return 0;
}

// bytes one 64-frame window of a launch takes in the slab pool: its float4 samples + (stream-pool kernel) the throughput-factor scratch.
// A launch's region = [samples of all its windows][scratch of all its windows].
static size_t sample_bytes_per_window(const crt_ctx* c, uint32_t passes) { return (size_t)c->tileCount * 256u * 64u * passes * 16u; }
static size_t window_bytes(const crt_ctx* c, uint32_t passes) { return sample_bytes_per_window(c, passes) + (c->usePool ? crt_pool_scratch_bytes_per_window(c->tileCount) : 0); }

// Latency mode: block tables of single-window launches from measured tile costs (100 MHz ticks of a tile's slowest wavefront).  A launch ends on its slowest
// wavefront, and a wavefront's speed is set by how many phases (NODE / TRI / SHADE) its lanes populate per trip, not by its lane count — so a tile given to
// 64 / L wavefronts of L lanes runs its streams' serial chains faster, at the price of 64 / L times the instruction issue.  Measured on the bunny's heaviest
// tiles (tools/latency_probe.py, everything else one wave per tile): time(L) / time(64) = 0.93 (32), 0.88 (16), 0.79 (8), 0.69 (4), 0.57 (2), 0.48 (1) — but the
// ratio differs from tile to tile (phase mix) and shrinks when the extra wavefronts compete for instruction issue, so the tables are found by FEEDBACK: each
// stage starts from the fastest stage so far (its lanes per tile and the costs measured under it), aims at `aim` x that stage's slowest tile, narrows every tile
// the model says would miss the aim and widens every tile that would make it with wavefronts twice as wide; crt_render times each stage and keeps the fastest.
// CRT_LAT_POLICY="share:lanes,.." replaces stage 1 by fixed steps (tiles costing >= share x the slowest get `lanes`) and stops there.
// one entry of a block table: local tile | first frame << 16 | log2(lanes) << 22 | window << 25 (render_tiles_kernel)
static uint32_t block_desc(uint32_t tile, uint32_t laneBase, uint32_t lanes, uint32_t window) { uint32_t lg = 0; while ((1u << lg) < lanes) lg++; return tile | (laneBase << 16) | (lg << 22) | (window << 25); }
// a block table in launch form: the blocks of wide wavefronts first (render_tiles_kernel), then those routed to render_narrow_kernel; both parts keep the table's
// order (most expensive tile first).  Returns the number of wide blocks.
// MEASURED, round 3 (tools/narrow_probe*.sh, profiles/r03_narrow_kernel.txt): render_narrow_kernel shortens a lone stream's chain — the heaviest bunny tile as 64
// one-lane wavefronts: 13.4 ms against 17 — but two-lane wavefronts are no faster than render_tiles_kernel's (18.8 ms: each ray waits for its neighbour's
// traversal), the chip holds fewer of its workgroups (32 KB of treetop in LDS each), and a settled single render is bound by machine time, not by its longest
// chain: with one-lane blocks routed to it the tuned 1280x720 / 64-spp render is 21.0 ms against 20.4 - 20.9, one rank's share of an 8-way tile split 16.3 ms
// against 17.3 - 18.3, and single stages of the tuner take 30 ms.  So nothing is routed to it unless a caller opts in (crt_config has no field for it: the
// environment variable CRT_NARROW_LANES = 1 | 2, read per launch, is the switch the tests and the probe tools use).
static uint32_t split_by_width(std::vector<uint32_t>& table)
{
    uint32_t maxNarrow = 0u;
    if (const char* e = hook("CRT_NARROW_LANES")) maxNarrow = std::min<uint32_t>((uint32_t)atoi(e), crt_narrow_max_lanes());
    if (maxNarrow == 0u) return (uint32_t)table.size();
    uint32_t limit = 3072u;                                        // 256 CUs x 12 wavefronts: all narrow blocks in flight at once, else the launch is bound by machine time
    if (const char* e = hook("CRT_NARROW_LIMIT")) limit = (uint32_t)atoi(e);
    std::vector<uint32_t> wide, narrow; wide.reserve(table.size()); narrow.reserve(table.size());
    for (uint32_t d : table) { if ((1u << ((d >> 22) & 7u)) <= maxNarrow) narrow.push_back(d); else wide.push_back(d); }
    if (narrow.size() > limit) return (uint32_t)table.size();
    const uint32_t nWide = (uint32_t)wide.size();
    table.swap(wide); table.insert(table.end(), narrow.begin(), narrow.end());
    return nWide;
}
static const uint32_t kLatLanes[7] = {64u, 32u, 16u, 8u, 4u, 2u, 1u};
static const double kLatG[7] = {1.0, 0.93, 0.88, 0.79, 0.69, 0.57, 0.48};
static int lat_index(uint32_t L) { int k = 0; while (k < 6 && kLatLanes[k] != L) k++; return k; }

// one feedback step of the latency tuner for one tile: it ran `lanes`-wide wavefronts and its slowest one took `cost`; the widest width whose modelled duration meets T
static uint8_t next_lanes(uint8_t lanes, uint32_t cost, double T)
{
    int k = lat_index(lanes); const double unit = (double)cost / kLatG[k];                     // the model's one-wave cost of this tile
    while (k < 6 && unit * kLatG[k] > T) k++;                                                  // narrower until the model meets the aim
    while (k > 0 && unit * kLatG[k - 1] <= 0.9 * T) k--;                                        // wider while twice as wide still makes it comfortably
    return (uint8_t)kLatLanes[k];
}
// tests (no GPU needed): next_lanes for n tiles
extern "C" int crt_debug_next_lanes(const uint8_t* lanes, const uint32_t* cost, uint32_t n, double T, uint8_t* out)
{
    if (!lanes || !cost || !out) return CRT_ERR_INVALID;
    for (uint32_t i = 0; i < n; i++) { if (lanes[i] == 0 || lanes[i] > 64 || (64 % lanes[i]) != 0) return CRT_ERR_INVALID; out[i] = next_lanes(lanes[i], cost[i], T); }
    return CRT_OK;
}

// The table of a stage, solved instead of stepped (round 3): every tile ran `lanes[i]`-wide wavefronts and its slowest took `cost[i]` (or: lanes 64 and the probe's estimate).
// A narrower table is a faster launch as long as the device holds ALL its wavefronts at once — a wavefront that has to wait for a slot starts its chain late — so the aim
// T is the LOWEST one whose table fits `budget` wavefronts (bisection; waves(T) falls as T rises), but not below what the most expensive tile takes as one-lane wavefronts,
// which bounds the launch anyway: cheaper tiles are not split to beat a time nothing can reach.  Returns T.
static double solve_block_table(const std::vector<uint8_t>& lanes, const std::vector<uint32_t>& cost, double budget, std::vector<uint8_t>& out)
{
    const size_t n = cost.size();
    double topUnit = 0; for (size_t i = 0; i < n; i++) topUnit = std::max(topUnit, (double)cost[i] / kLatG[lat_index(lanes[i])]);
    out.assign(n, 64);
    if (topUnit <= 0) return 0;
    auto waves = [&](double T) { double w = 0; for (size_t i = 0; i < n; i++) w += 64.0 / next_lanes(lanes[i], cost[i], T); return w; };
    double lo = kLatG[6] * topUnit, hi = topUnit;
    if (waves(lo) > budget) { for (int it = 0; it < 14; it++) { const double mid = 0.5 * (lo + hi); if (waves(mid) <= budget) hi = mid; else lo = mid; } lo = hi; }
    for (size_t i = 0; i < n; i++) out[i] = next_lanes(lanes[i], cost[i], lo);
    return lo;
}
// tests (no GPU needed): the solved table for n tiles and a wavefront budget; returns the aim through *T
extern "C" int crt_debug_solve_block_table(const uint8_t* lanes, const uint32_t* cost, uint32_t n, double budget, uint8_t* out, double* T)
{
    if (!lanes || !cost || !out) return CRT_ERR_INVALID;
    for (uint32_t i = 0; i < n; i++) if (lanes[i] == 0 || lanes[i] > 64 || (64 % lanes[i]) != 0) return CRT_ERR_INVALID;
    std::vector<uint8_t> L; const double t = solve_block_table(std::vector<uint8_t>(lanes, lanes + n), std::vector<uint32_t>(cost, cost + n), budget, L);
    if (n) memcpy(out, L.data(), n);
    if (T) *T = t;
    return CRT_OK;
}
static double lat_budget(crt_ctx* c)
{
    if (!c->latSlots) c->latSlots = crt_render_resident_waves(c->cfg.device, c->hScene.kind, c->ldsBytes);
    double share = 0.98;
    if (const char* e = hook("CRT_LAT_BUDGET")) { const double v = atof(e); if (v > 0) share = v; }
    return share * (double)c->latSlots;
}

static int upload_block_table(crt_ctx* c, const std::vector<uint8_t>& lanes, const std::vector<uint32_t>& cost)
{
    const uint32_t n = c->tileCount;
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });      // issued most expensive tile first
    std::vector<uint32_t> table; table.reserve((size_t)n * 2);
    for (uint32_t r = 0; r < n; r++) {
        const uint32_t tl = order[r], L = lanes[tl];
        for (uint32_t base = 0; base < 64u; base += L) table.push_back(block_desc(tl, base, L, 0u));
    }
    if (table.size() > 0x7fffffffull) return c->fail(CRT_ERR_INVALID, "block table too large");
    c->nBlocksWide = split_by_width(table);
    if (hook("CRT_LAT_VERBOSE")) fprintf(stderr, "[crt] latency table: %zu wavefronts for %u tiles (slowest tile of the base stage %.2f ms)\n", table.size(), n, n ? cost[order[0]] * 1e-5 : 0.0);
    if (c->descCap < table.size()) {
        if (c->dBlockDesc) (void)hipFree(c->dBlockDesc);
        if (c->hBlockDesc) (void)hipHostFree(c->hBlockDesc);
        c->dBlockDesc = nullptr; c->hBlockDesc = nullptr; c->descCap = 0;
        const size_t cap = table.size() + table.size() / 2;
        HIPCK(c, hipMalloc((void**)&c->dBlockDesc, cap * 4));
        HIPCK(c, hipHostMalloc((void**)&c->hBlockDesc, cap * 4, hipHostMallocDefault));
        c->descCap = (uint32_t)cap;
    } else if (c->descReady) HIPCK(c, hipEventSynchronize(c->descReady));     // the staging buffer's previous copy (long done)
    if (!c->descReady) HIPCK(c, hipEventCreateWithFlags(&c->descReady, hipEventDisableTiming));
    memcpy(c->hBlockDesc, table.data(), table.size() * 4);
    // on the main stream: ordered behind every launch submitted so far (each launch's accumulate waits for it there), so the previous table is no longer read
    HIPCK(c, hipMemcpyAsync(c->dBlockDesc, c->hBlockDesc, table.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipEventRecord(c->descReady, c->stream));
    c->nBlocks = (uint32_t)table.size();
    return 0;
}

// the costs of stage `c->costStage` have arrived in hTileCost: remember them, then build and install the next stage's table (or, after the last stage, the fastest one's)
static int next_block_table(crt_ctx* c)
{
    const uint32_t n = c->tileCount; const int K = crt_ctx::kLatStages;
    const int s = c->costStage;
    if (s == 0 && !c->latWarm && !c->latProbed) { c->latWarm = true; return 0; }          // measure the one-wave launch once more, warm (a probed stage 0 is not the base of anything: every stage is solved from the one before it)
    c->latCost[s].assign(c->hTileCost, c->hTileCost + n);
    if (s == 0 && !c->latProbed) c->latL[0].assign(n, 64);
    if (c->tuneCount[s] && (c->latBest == s || !c->tuneCount[c->latBest] || c->tuneMs[s] < c->tuneMs[c->latBest])) c->latBest = s;
    bool last = s >= K;
    std::vector<std::pair<double, uint32_t>> steps;
    if (const char* e = hook("CRT_LAT_POLICY")) {
        for (const char* p = e; *p;) {
            char* q = nullptr; const double sh = strtod(p, &q); if (q == p || *q != ':') break;
            const long L = strtol(q + 1, &q, 10); if (L < 1 || L > 64 || (64 % L) != 0) break;
            steps.push_back({sh, (uint32_t)L});
            p = (*q == ',') ? q + 1 : q; if (*q != ',') break;
        }
        if (s >= 1) last = true;
    }
    auto install = [&](int stage) -> int {                                  // the table of an earlier stage back onto the device (an unprobed stage 0 needs none)
        if ((stage != 0 || c->latProbed) && stage != c->latStage) { const int r = upload_block_table(c, c->latL[stage], c->latCost[stage]); if (r) return r; }
        c->latStage = stage; return 0;
    };
    auto fastest = [&](int except) { int b = -1; for (int k = 0; k <= K; k++) if (k != except && c->tuneCount[k] && (b < 0 || c->tuneMs[k] < c->tuneMs[b])) b = k; return b; };
    auto finish = [&]() -> int {
        c->latDone = true; c->latConfirming = false;
        const int b = fastest(-1); c->latBest = b < 0 ? 0 : b;
        if (hook("CRT_LAT_FORCE")) c->latBest = s;                       // diagnostics: keep the last table whatever its time
        if (hook("CRT_LAT_VERBOSE")) { fprintf(stderr, "[crt] latency stages:"); for (int k = 0; k <= K; k++) if (c->tuneCount[k]) fprintf(stderr, " %d: %.2f ms", k, c->tuneMs[k]); fprintf(stderr, " -> %d\n", c->latBest); }
        return install(c->latBest);
    };
    if (c->latConfirming) {                                               // a confirmation launch of stage s has been timed (harvest_tuning keeps each stage's minimum)
        if (!c->latQueue.empty()) c->latQueue.erase(c->latQueue.begin());
        if (!c->latQueue.empty()) return install(c->latQueue.front());
        return finish();
    }
    if (last) {
        // every stage has ONE timing so far, and launch durations scatter by a few per cent: the two fastest stages run once more before the choice is final
        const int a = fastest(-1), b2 = fastest(a);
        if (hook("CRT_LAT_FORCE") || a < 0 || b2 < 0) return finish();
        c->latQueue = {a, b2}; c->latConfirming = true;
        return install(a);
    }
    const int b = c->latBest;                                              // base: the fastest stage so far
    const std::vector<uint8_t>& baseL = c->latL[b]; const std::vector<uint32_t>& baseC = c->latCost[b];
    uint32_t top = 0; for (uint32_t i = 0; i < n; i++) top = baseC[i] > top ? baseC[i] : top;
    std::vector<uint8_t>& L = c->latL[s + 1]; L = baseL;
    if (!steps.empty()) {
        for (uint32_t i = 0; i < n; i++)
            for (const auto& st : steps) if (top > 0 && (double)baseC[i] >= st.first * (double)top) { L[i] = (uint8_t)st.second; break; }
    } else {
        if (const char* e = hook("CRT_LAT_AIM")) {                          // diagnostics: the stepping tuner of round 2 (aim = a fraction of the base stage's slowest tile)
            double aim = atof(e); if (!(aim > 0)) aim = (s == 0 && !c->latProbed) ? 0.64 : 0.92;
            for (uint32_t i = 0; i < n; i++) L[i] = next_lanes(baseL[i], baseC[i], aim * (double)top);
        } else {
            // solved from the LATEST stage's measurement (each tile's cost at the width it just ran is the best estimate of its one-wave cost there is, whether or not
            // that stage was the fastest); keeping the fastest stage guards the choice
            const double T = solve_block_table(c->latL[s], c->latCost[s], lat_budget(c), L);
            if (hook("CRT_LAT_VERBOSE")) fprintf(stderr, "[crt] stage %d: aim %.2f ms\n", s + 1, T * 1e-5);
        }
    }
    const int r = upload_block_table(c, L, (steps.empty() && !hook("CRT_LAT_AIM")) ? c->latCost[s] : baseC);
    if (r) return r;
    c->latStage = s + 1;
    return 0;
}

// Timing pairs of launches that have completed are folded into running totals and recycled, so a host that renders forever and never
// asks for the timing (an interactive Tick loop) keeps a bounded number of HIP events alive.
// latency-mode auto-tuning: durations of completed single-window launches, by mode (each launch is looked at once)
static void harvest_tuning(crt_ctx* c)
{
    for (auto& ev : c->evRender) {
        if (ev.mode < 0 || ev.seen) continue;
        if (hipEventQuery(ev.b) != hipSuccess) break;                   // launches complete in order per stream; stop at the first unfinished one
        float t = 0;
        if (hipEventElapsedTime(&t, ev.a, ev.b) == hipSuccess) {
            if (ev.mode >= 100) { double& m = c->jobTrialMs[ev.mode - 100]; m = (m > 0 && m < t) ? m : t; }      // a job of the current plan's shape: plain (100) / planned (101)
            else { c->tuneMs[ev.mode] = c->tuneCount[ev.mode] ? (c->tuneMs[ev.mode] < t ? c->tuneMs[ev.mode] : t) : t; c->tuneCount[ev.mode]++; }
        }
        ev.seen = true;
    }
}

static void fold_completed(crt_ctx* c, std::deque<EventPair>& list, double* ms, uint32_t* count)
{
    while (list.size() > 64 && hipEventQuery(list.front().b) == hipSuccess) {
        float t = 0;
        if (hipEventElapsedTime(&t, list.front().a, list.front().b) == hipSuccess) *ms += t;
        if (count) (*count)++;
        c->evPool.push_back(list.front()); list.pop_front();
    }
}

static int take_event(crt_ctx* c, std::deque<EventPair>& list, EventPair* out)
{
    EventPair ev;
    if (c->evPool.empty()) {
        HIPCK(c, hipEventCreate(&ev.a)); HIPCK(c, hipEventCreate(&ev.b));
    } else { ev = c->evPool.back(); c->evPool.pop_back(); }
    ev.mode = -1; ev.seen = false;
    list.push_back(ev); *out = ev;
    return 0;
}

// Sizes the sample-slab pool for a crt_render of `frames` frames and returns the frames per launch to use (*maxFOut).
// One launch renders up to cfg.maxFramesPerLaunch frames = that many / 64 windows; its samples need windowBytes per window.  The pool
// grows to the high-water mark only (hipMalloc synchronises the device and takes seconds for tens of GB): room for the largest
// launch — for two of them when a call needs several launches — and for at least eight windows (consecutive single-window calls
// overlap on the streams), never more than half of the free HBM.
static int ensure_pool(crt_ctx* c, uint32_t frames, uint32_t passes, uint32_t* maxFOut)
{
    const size_t windowBytes = window_bytes(c, passes);
    uint32_t maxF = (uint32_t)c->cfg.maxFramesPerLaunch;
    uint32_t maxW = maxF >= 64u ? maxF / 64u : 1u;
    const uint32_t wantW = (frames + 63u) / 64u;                                          // (an upper bound when maxF < 64)
    const uint32_t callW = wantW < maxW ? wantW : maxW;                                   // windows of this call's largest launch
    size_t want = (size_t)callW * windowBytes * (wantW > callW ? 2u : 1u);
    if (want < 8 * windowBytes) want = 8 * windowBytes;
    if (want > c->poolBytes && !(c->poolCapped && windowBytes <= c->poolBytes)) {
        HIPCK(c, hipStreamSynchronize(c->stream));
        for (auto st : c->streams) HIPCK(c, hipStreamSynchronize(st));
        for (auto& r : c->inflight) c->freeEvents.push_back(r.freed);
        c->inflight.clear(); c->poolHead = 0;
        if (c->pool) { HIPCK(c, hipFree(c->pool)); c->pool = nullptr; c->poolBytes = 0; }
        size_t freeB = 0, totalB = 0;
        HIPCK(c, hipMemGetInfo(&freeB, &totalB));
        const size_t budget = freeB / 2;
        if (budget < windowBytes) return c->fail(CRT_ERR_DEVICE, "not enough free HBM for one 64-frame sample slab (%zu bytes needed, %zu free)", windowBytes, freeB);
        c->poolCapped = want > budget;
        if (want > budget) want = budget;
        want = want / windowBytes * windowBytes;
        HIPCK(c, hipMalloc((void**)&c->pool, want));
        c->poolBytes = want;
    }
    {   // launches must fit the pool; when the call needs several launches leave room for two in flight
        uint32_t fitW = (uint32_t)(c->poolBytes / windowBytes);
        if (wantW > fitW && fitW >= 2) fitW /= 2;
        if (maxW > fitW) maxW = fitW;
    }
    if (maxF >= 64u) maxF = maxW * 64u;
    *maxFOut = maxF;
    return 0;
}

int crt_reserve(crt_ctx* c, uint32_t frames, uint32_t passes)
{
    if (!c) return CRT_ERR_INVALID;
    if (passes < 1 || passes > 4) return c->fail(CRT_ERR_INVALID, "passes must be 1..4");
    HIPCK(c, hipSetDevice(c->cfg.device));
    if (c->tileCount == 0 || frames == 0) return CRT_OK;
    uint32_t maxF = 0;
    return ensure_pool(c, frames, passes, &maxF);
}

// a region of `need` bytes of the slab pool for a launch on stream `st`; regions still in use are waited for on the GPU (never on the host)
static int take_region(crt_ctx* c, size_t need, hipStream_t st, size_t* offOut)
{
    for (;;) {
        bool ok = false; size_t off = 0;
        if (c->inflight.empty()) { off = 0; ok = need <= c->poolBytes; }
        else {
            const size_t tail = c->inflight.front().off;                     // oldest region still held
            if (c->poolHead > tail) {                                        // free: [head, end) and [0, tail)
                if (c->poolBytes - c->poolHead >= need) { off = c->poolHead; ok = true; }
                else if (tail >= need) { off = 0; ok = true; }
            } else if (c->poolHead < tail && tail - c->poolHead >= need) { off = c->poolHead; ok = true; }
        }
        if (ok) {
            // every launch is ordered behind the release of ALL space handed out again so far (releases are ordered on the main stream)
            if (c->mustWait) HIPCK(c, hipStreamWaitEvent(st, c->mustWait, 0));
            c->poolHead = off + need; *offOut = off;
            return 0;
        }
        if (c->inflight.empty()) return c->fail(CRT_ERR_DEVICE, "slab pool of %zu bytes cannot hold a launch of %zu bytes", c->poolBytes, need);
        if (c->mustWait) c->freeEvents.push_back(c->mustWait);
        c->mustWait = c->inflight.front().freed;
        c->inflight.pop_front();
    }
}

// the job-cost buffer: one uint32 per tile, then (8-byte aligned) three uint64: ~(first wavefront's start clock), last wavefront's start clock, wave time after it
static size_t job_cost_clk_offset(const crt_ctx* c) { return (((size_t)c->tileCount + 1u) & ~(size_t)1u) * 4u; }
static size_t job_cost_bytes(const crt_ctx* c) { return job_cost_clk_offset(c) + 24u; }

// A job's tile costs have arrived (hJobCost): dispatch order = most expensive tile first from now on; and what ONE window of this image costs the machine under
// the pool — the planner's yardstick.  The sum of the wavefront durations is no measure of it (waves that share a SIMD with four others last longer; the
// expensive tiles' wavefronts, which outlive the crowd, do not), so it comes from the launch's dispatch: a launch that oversubscribes the chip keeps every wavefront
// slot busy until its last wavefront starts (S0 = last start - first start), and drains afterwards — the wave time spent after S0, summed by the wavefronts
// themselves, divided by the slots.  One-stream-per-lane launches need 1.2x the pool's machine time (4.24 against 3.52 ms per window on the bunny).
static int adopt_job_costs(crt_ctx* c)
{
    const uint32_t n = c->tileCount;
    c->jobCost.assign(c->hJobCost, c->hJobCost + n);
    {
        const unsigned long long* clk = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(c->hJobCost) + job_cost_clk_offset(c));
        const double s0 = (clk[0] && clk[1] && clk[1] > ~clk[0]) ? (double)(clk[1] - ~clk[0]) : 0.0;
        const double drain = (double)clk[2] / (double)(c->recResident ? c->recResident : 1u);
        c->poolWindowTicks = 0;
        if (c->recWindows && (double)c->recWaves >= 1.5 * (double)c->recResident && s0 > 0) c->poolWindowTicks = (s0 + drain) / (double)c->recWindows / (c->recPool ? 1.0 : 1.2);
        if (hook("CRT_LAT_VERBOSE")) fprintf(stderr, "[crt] measured %u windows with %s: %u wavefronts, last one started %.2f ms after the first, then %.2f ms of drain -> %.3f ms of machine time per window under the pool\n", c->recWindows, c->recPool ? "the pool" : "one stream per lane", c->recWaves, s0 * 1e-5, drain * 1e-5, c->poolWindowTicks * 1e-5);
    }
    c->jobOrder.resize(n);
    for (uint32_t i = 0; i < n; i++) c->jobOrder[i] = i;
    std::stable_sort(c->jobOrder.begin(), c->jobOrder.end(), [&](uint32_t a, uint32_t b) { return c->jobCost[a] > c->jobCost[b]; });
    c->jobCostValid = true; c->planValid = false;
    return upload_tile_order(c, c->jobOrder);
}

// Plan of a job (one launch of `windows` windows) once the tile costs are known.  A launch ends on its slowest wavefront, and how long a wavefront runs is set by
// the serial chains of its streams: per tile and window, a stream-pool wavefront (128 streams on 64 lanes; the fewest instructions per sample) runs 2.4x as long
// as a one-stream-per-lane wavefront of render_tiles_kernel, and that one can be cut further by handing the tile's 64 streams to 64 / L wavefronts of L lanes
// (kLatG; at 64 / L times the instruction issue).  A job many times longer than its slowest pool wavefront should be all pool; a short one (few windows, or one
// rank's share of a multi-GPU tile split) ends on those wavefronts with the chip nearly empty (tools/pool_timeline.py: 20 windows of the bunny, 4 096 waves in
// flight until 65 ms, then a tail to 83 ms).  The plan is the smallest makespan T for which every tile can take the CHEAPEST class whose wavefronts last <= T
//     pool (2.4 c, if this launch may use the pool) | one wavefront per window (1.1 c under load) | 64 / L wavefronts per window (g(L) c)
// and the machine time of all of them (wave time / wavefronts the chip advances at once) still fits T.  Tiles are sorted by cost, so the classes are contiguous:
// the first `head` tiles of the order go to a block table (render_tiles_kernel, dispatched first), the rest to the pool launch — or, in a launch without the pool,
// everything goes to the table as soon as one tile needs narrow wavefronts.  Returns the table's blocks in `table` (empty: no table needed).
static void plan_job(const crt_ctx* c, uint32_t windows, uint32_t frames, bool pool, std::vector<uint32_t>& table, uint32_t* head)
{
    table.clear(); *head = 0;
    const uint32_t n = c->tileCount;
    if (!c->jobCostValid || windows < 2u || windows > 64u || n > 0x10000u || c->cfg.collectStats || hook("CRT_SPLIT_OFF")) return;
    if (pool && c->streams.size() < 2) return;
    if (const char* e = hook("CRT_SPLIT_FORCE")) {                      // tests: the first h tiles through the table, alternating wavefront widths
        const uint32_t h = std::min<uint32_t>((uint32_t)atoi(e), pool ? n - 1u : n);
        static const uint32_t Ls[4] = {64u, 16u, 2u, 1u};
        for (uint32_t r = 0; r < (pool ? h : n); r++) { const uint32_t L = r < h ? Ls[r & 3u] : 64u; for (uint32_t w = 0; w < windows; w++) for (uint32_t b0 = 0; b0 < 64u; b0 += L) table.push_back(block_desc(c->jobOrder[r], b0, L, w)); }
        *head = h; return;
    }
    // cost unit: a pool wavefront's duration per 64 streams while the costs were measured (render_pool_kernel).  In a saturated job the wavefronts of the most expensive
    // tiles run 1.2x longer than that (82 ms against 68 ms on the bunny); one-stream-per-lane wavefronts need 1.2x the machine time of the pool's.
    double poolLong = 2.4, wideLoad = 1.1, poolSlots = 4096.0, wideMt = 1.2, narrowSlots = 3500.0;
    if (const char* e = hook("CRT_PLAN_NARROW_SLOTS")) narrowSlots = atof(e);
    const double K = (double)windows;
    // machine time of one window's pool wavefront of a tile, per unit of its cost: from the measured machine time per window when the measuring launch
    // oversubscribed the chip (adopt_job_costs), else from the wavefront durations (which then ran without much competition)
    double costSum = 0; for (uint32_t v : c->jobCost) costSum += (double)v;
    const double perCost = (c->poolWindowTicks > 0 && costSum > 0) ? c->poolWindowTicks / costSum : 1.0 / poolSlots;
    // machine time (ticks) of tile cost v in the cheapest class that lasts <= T; cls: 0 pool, 1 wide, 2 + k narrow (kLatLanes[k])
    auto cheapest = [&](double v, double T, int* cls) -> double {
        if (pool && poolLong * v <= T) { *cls = 0; return K * v * perCost; }
        if (wideLoad * v <= T) { *cls = 1; return K * wideMt * v * perCost; }
        int k = 1; while (k < 6 && kLatG[k] * v > T) k++;
        *cls = 2 + k; return K * (64.0 / kLatLanes[k]) * kLatG[k] * v / narrowSlots;
    };
    auto machine = [&](double T) { double m = 0; int cls; for (uint32_t v : c->jobCost) m += cheapest((double)v, T, &cls); return m; };
    const double top = (double)c->jobCost[c->jobOrder[0]];
    double lo = top * kLatG[6], hi = std::max(poolLong * top, machine(1e30)) * 1.01;
    if (machine(lo) <= lo) hi = lo;
    else for (int it = 0; it < 50; it++) { const double mid = 0.5 * (lo + hi); if (machine(mid) <= mid) hi = mid; else lo = mid; }
    const double T = hi;
    uint32_t h = 0, narrow = 0; std::vector<uint8_t> lanes(n, 64);
    for (uint32_t r = 0; r < n; r++) {
        int cls; (void)cheapest((double)c->jobCost[c->jobOrder[r]], T, &cls);
        if (cls != 0) h = r + 1u;
        if (cls >= 2) { lanes[r] = (uint8_t)kLatLanes[cls - 2]; narrow++; }
    }
    if (pool && h >= n) h = n - 1u;
    if (hook("CRT_LAT_VERBOSE")) fprintf(stderr, "[crt] job of %u windows, %u tiles: makespan aim %.1f ms (most expensive tile %.2f ms); %u tiles to the block table, %u of them narrow, %s\n", windows, n, T * 1e-5, top * 1e-5, pool ? h : (narrow ? n : 0u), narrow, pool ? "rest to the pool" : "no pool");
    if (pool ? h == 0u : narrow == 0u) return;
    const uint32_t upto = pool ? h : n;
    for (uint32_t r = 0; r < upto; r++) {
        const uint32_t L = lanes[r];
        for (uint32_t w = 0; w < windows; w++) {
            const uint32_t fw = std::min(64u, frames - w * 64u);               // frames of this window: no wavefronts for frames past the end
            for (uint32_t b0 = 0; b0 < fw; b0 += L) table.push_back(block_desc(c->jobOrder[r], b0, L, w));
        }
    }
    if (table.size() > (4u << 20)) { table.clear(); *head = 0; return; }          // (never seen: the machine-time bound keeps narrow tiles few; a plain launch is always valid)
    *head = pool ? h : 0u;
}

// ... and its table on the device (cached for launches of the same shape)
static int install_job_plan(crt_ctx* c, uint32_t windows, uint32_t frames, bool pool)
{
    if (c->planValid && c->planWindows == windows && c->planFrames == frames && c->planPool == pool) return 0;
    std::vector<uint32_t> table; uint32_t head = 0;
    plan_job(c, windows, frames, pool, table, &head);
    c->jobBlocksWide = split_by_width(table);
    c->planValid = true; c->planWindows = windows; c->planFrames = frames; c->planPool = pool; c->jobHead = head; c->jobBlocks = (uint32_t)table.size();
    c->jobTrialMs[0] = c->jobTrialMs[1] = 0;
    if (table.empty()) return 0;
    if (c->jobDescCap < table.size()) {
        if (c->dJobDesc) (void)hipFree(c->dJobDesc);
        if (c->hJobDesc) (void)hipHostFree(c->hJobDesc);
        c->dJobDesc = nullptr; c->hJobDesc = nullptr; c->jobDescCap = 0;
        const size_t cap = table.size() + table.size() / 2;
        HIPCK(c, hipMalloc((void**)&c->dJobDesc, cap * 4));
        HIPCK(c, hipHostMalloc((void**)&c->hJobDesc, cap * 4, hipHostMallocDefault));
        c->jobDescCap = (uint32_t)cap;
    } else if (c->jobDescReady) HIPCK(c, hipEventSynchronize(c->jobDescReady));
    if (!c->jobDescReady) HIPCK(c, hipEventCreateWithFlags(&c->jobDescReady, hipEventDisableTiming));
    memcpy(c->hJobDesc, table.data(), table.size() * 4);
    HIPCK(c, hipMemcpyAsync(c->dJobDesc, c->hJobDesc, table.size() * 4, hipMemcpyHostToDevice, c->stream));      // main stream: behind every launch submitted so far
    HIPCK(c, hipEventRecord(c->jobDescReady, c->stream));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// One render launch of crt_render = the three steps below, each owning its part of the context:
//   tuner_prepare     single-window launches (8 .. 64 frames): the LATENCY tuner — owns lat*, cost*, dBlockDesc / nBlocks* (probe, stages, confirmation: next_block_table)
//   planner_prepare   jobs (> 64 frames): tile-cost measurement and the PLANNER — owns jobCost*, rec*, plan*, dJobDesc / jobBlocks*, jobTrial* (plan_job, install_job_plan)
//   launch_render_kernels   the LAUNCHER: render_pool_kernel / render_tiles_kernel / render_narrow_kernel on their streams, joined before the launch's end event
// ---------------------------------------------------------------------------------------------------------------------------------------------
struct Launch {
    hipStream_t st = nullptr; EventPair ev{}; void* slab = nullptr; uint32_t sppFirst = 0, nf = 0, passes = 1, windows = 1;
    const uint32_t* blockDesc = nullptr; uint32_t nBlocks = 0, nBlocksWide = 0;      // the latency mode's table (single-window launch), or none
    bool wantCost = false, wantJobCost = false, pool = false;
    uint32_t head = 0, jobBlocks = 0, jobBlocksWide = 0; unsigned long long* jobClk = nullptr;      // a planned job: table blocks + first tile rank of the pool launch
};

static int next_block_table(crt_ctx* c);
static int probe_tile_costs(crt_ctx* c, hipStream_t st);

static int tuner_prepare(crt_ctx* c, Launch& L)
{
    int r;
    // ... only when the GPU is idle at submission: a caller that queues launch after launch wants throughput, and narrow wavefronts buy latency with issue
    // slots (56 queued single-window calls: 6.4 ms each with one wave per tile, 12.4 ms with the tuned table)
    const bool gpuIdle = !c->lastRenderEnd || hipEventQuery(c->lastRenderEnd) == hipSuccess;
    if (L.nf <= 64u && L.nf >= 8u && gpuIdle && !c->cfg.collectStats && c->tileCount <= 0x10000u && !hook("CRT_LAT_OFF")) {
        if (c->costPending && hipEventQuery(c->costCopied) == hipSuccess) { c->costPending = false; harvest_tuning(c); if ((r = next_block_table(c))) return r; }
        if (c->latStage == 0 && !c->latProbed && !c->latWarm && !c->costPending && L.nf == 64u && !hook("CRT_LAT_POLICY") && !hook("CRT_LAT_NO_PROBE")) { if ((r = probe_tile_costs(c, L.st))) return r; }
        const int stage = c->latStage;
        if (stage || c->latProbed) { L.blockDesc = c->dBlockDesc; L.nBlocks = c->nBlocks; L.nBlocksWide = c->nBlocksWide; HIPCK(c, hipStreamWaitEvent(L.st, c->descReady, 0)); }
        L.wantCost = (!c->latDone && !c->costPending) || (c->latDone && hook("CRT_LAT_RECORD"));      // (the latter: diagnostics, crt_debug_tile_costs)
        c->evRender.back().mode = c->latDone ? -1 : stage;
        if (L.wantCost && !c->latDone) c->costStage = stage;
    }
    if (L.wantCost) {
        if (!c->dTileCost) {
            HIPCK(c, hipMalloc((void**)&c->dTileCost, (size_t)c->tileCount * 4));
            HIPCK(c, hipHostMalloc((void**)&c->hTileCost, (size_t)c->tileCount * 4, hipHostMallocDefault));
            HIPCK(c, hipEventCreateWithFlags(&c->costCopied, hipEventDisableTiming));
        }
        else HIPCK(c, hipStreamWaitEvent(L.st, c->costCopied, 0));          // behind an earlier measurement (possibly on another stream) that a camera change abandoned
        HIPCK(c, hipMemsetAsync(c->dTileCost, 0, (size_t)c->tileCount * 4, L.st));
    }
    return 0;
}

static int planner_prepare(crt_ctx* c, Launch& L)
{
    int r;
    // jobs: measure the tile costs once per camera / scene (first job launch), adopt them when they have arrived
    if (L.nf > 64u && !c->cfg.collectStats) {
        if (c->jobCostPending && hipEventQuery(c->jobCostCopied) == hipSuccess) { c->jobCostPending = false; if ((r = adopt_job_costs(c))) return r; HIPCK(c, hipStreamWaitEvent(L.st, c->orderReady, 0)); }
        L.wantJobCost = !c->jobCostValid && !c->jobCostPending && L.nf >= 128u;       // (a pool launch records full 128-stream wavefronts only)
        if (L.wantJobCost) {
            if (!c->dJobCost) {
                HIPCK(c, hipMalloc((void**)&c->dJobCost, job_cost_bytes(c)));
                HIPCK(c, hipHostMalloc((void**)&c->hJobCost, job_cost_bytes(c), hipHostMallocDefault));
                HIPCK(c, hipEventCreateWithFlags(&c->jobCostCopied, hipEventDisableTiming));
            } else HIPCK(c, hipStreamWaitEvent(L.st, c->jobCostCopied, 0));      // behind an earlier measurement that a camera change abandoned
            HIPCK(c, hipMemsetAsync(c->dJobCost, 0, job_cost_bytes(c), L.st));
        }
    }
    // Which render kernel: the stream pool executes a third fewer instructions per sample, but its wavefronts own 128 streams for 64 lanes, so the most
    // expensive tiles take about twice as long per wavefront; a launch that is not many times larger than the machine (4 096 - 5 120 wavefronts in
    // flight) ends on those and is faster with one stream per lane.  Measured cross-over (tools/crossover.py, bench.py --steps): bunny 1280x720 at 17 - 20
    // windows (20 windows: 81.0 ms pool, 87.6 ms tiles), TLAS scene at ~28, watch-tower 1920x1080 at 7 — 56 000 ... 100 000 (tile, window) pairs; the
    // threshold sits at the low end of that range.
    // With the tile costs known (most expensive first + split, see plan_job) the pool is never slower than one stream per lane from ~12 windows of 720p on
    // (tools/split_probe.py: bunny 14 windows 57.8 against 59.6 ms, two-level scene 16 windows 79 against 94 ms, watch-tower 1080p 7 windows 155.6 against 161.1 ms).
    const uint64_t minWaves = c->poolMinWaves != 65000u ? c->poolMinWaves : (c->jobCostValid ? 43000u : 65000u);
    L.pool = c->usePool && c->hScene.ref16ok && (uint64_t)c->tileCount * L.windows >= minWaves && (c->poolMinWaves == 0 || L.nf > 64u);
    if (L.nf > 64u && c->jobCostValid) {
        if ((r = install_job_plan(c, L.windows, L.nf, L.pool))) return r;
        // The plan is a MODEL (machine time within ~15 %): for a job shape that repeats (a progressive render) it is checked by measurement — the first launch of the
        // shape runs planned, the second plain (same kernel choice, cost-ordered, no table), and whichever was faster is kept: "planned is never slower than the
        // plain launch" holds by construction from the third launch on (tags 100 / 101 of the timing pairs, harvest_tuning).
        int use = 1;                                                           // 1 planned, 0 plain
        if (c->jobBlocks && !hook("CRT_PLAN_NO_TRIAL")) {
            if (c->jobTrialMs[1] > 0 && c->jobTrialMs[0] > 0) use = c->jobTrialMs[1] <= c->jobTrialMs[0] ? 1 : 0;
            else if (c->jobTrialMs[1] > 0 && c->jobTrialMs[0] == 0) use = 0;   // the planned launch has been timed: time the plain one
            c->evRender.back().mode = 100 + use;
        }
        if (use) { L.head = c->jobHead; L.jobBlocks = c->jobBlocks; L.jobBlocksWide = c->jobBlocksWide; }
    }
    if (L.jobBlocks) HIPCK(c, hipStreamWaitEvent(L.st, c->jobDescReady, 0));
    if (L.wantJobCost) {                                                 // what adopt_job_costs needs to know about the measuring launch
        L.jobClk = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->dJobCost) + job_cost_clk_offset(c));
        c->recPool = L.pool; c->recWindows = L.windows;
        c->recWaves = L.pool ? c->tileCount * ((L.nf + crt_pool_streams(L.nf) - 1u) / crt_pool_streams(L.nf)) : c->tileCount * L.windows;
        c->recResident = L.pool ? 4096u : 5120u;                            // 4 / 5 wavefronts per SIMD (render_pool_kernel / render_tiles_kernel)
    }
    return 0;
}

static hipError_t launch_render_kernels(crt_ctx* c, const Launch& L)
{
    hipStream_t st = L.st;
    hipError_t le = hipSuccess;
    // a second kernel of the same launch on another stream: released by the launch's start event, joined before its end event
    auto join = [&](hipStream_t other) -> hipError_t {
        if (other == st) return hipSuccess;
        hipError_t e = hipSuccess;
        if (c->splitEvents.size() < 32) { hipEvent_t ne; e = hipEventCreateWithFlags(&ne, hipEventDisableTiming); if (e == hipSuccess) c->splitEvents.push_back(ne); }
        if (e == hipSuccess) { hipEvent_t je = c->splitEvents[c->splitSeq++ % c->splitEvents.size()]; e = hipEventRecord(je, other); if (e == hipSuccess) e = hipStreamWaitEvent(st, je, 0); }
        return e;
    };
    // a block table = [wide blocks: render_tiles_kernel on this launch's stream][blocks routed to render_narrow_kernel (split_by_width: none unless opted in) on the
    // highest-priority stream, submitted first — they are the most expensive tiles' and the launch ends on them]
    auto launch_table = [&](const uint32_t* desc, uint32_t nAll, uint32_t nWide, uint32_t* cost) -> hipError_t {
        hipError_t e = hipSuccess;
        hipStream_t sn = st;
        if (nAll > nWide) {
            if (!c->narrowStream) { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); e = hipStreamCreateWithPriority(&c->narrowStream, hipStreamNonBlocking, hi); if (e != hipSuccess) return e; }
            sn = c->narrowStream;
            e = hipStreamWaitEvent(sn, L.ev.a, 0);
            if (e == hipSuccess) e = crt_launch_render_narrow(&c->hScene, L.slab, c->dCounters, c->tileFirst, c->tileStride, c->tileCount, (uint32_t)c->tilesX,
                                                              L.sppFirst, L.nf, L.passes, c->ldsBytes, desc + nWide, nAll - nWide, cost, sn);
        }
        if (e == hipSuccess && nWide)
            e = crt_launch_render(&c->hScene, L.slab, c->dCounters, c->dTileClocks, c->dTileOrder, c->tileFirst, c->tileStride, c->tileCount, (uint32_t)c->tilesX,
                                  L.sppFirst, L.nf, L.passes, c->ldsBytes, 0, desc, nWide, cost, 0u, nullptr, st);
        if (e == hipSuccess) e = join(sn);                                 // (after both are submitted: the two kernels run side by side)
        return e;
    };
    if (L.pool) {
        void* scratch = (char*)L.slab + (size_t)L.windows * sample_bytes_per_window(c, L.passes);
        hipStream_t st2 = st;
        if (L.jobBlocks) {
            // the expensive tiles first, through the block table, on this launch's stream; the pool for the rest on the next stream, released by the same start event
            le = launch_table(c->dJobDesc, L.jobBlocks, L.jobBlocksWide, nullptr);
            st2 = c->streams[(size_t)(c->launchSeq++ % c->streams.size())];
            if (le == hipSuccess && st2 != st) le = hipStreamWaitEvent(st2, L.ev.a, 0);
        }
        if (le == hipSuccess)
            le = crt_launch_render_pool(&c->hScene, L.slab, scratch, c->dCounters, c->dTileClocks, c->dTileOrder, c->tileFirst, c->tileStride, c->tileCount, (uint32_t)c->tilesX,
                                        L.sppFirst, L.nf, L.passes, c->cfg.collectStats, L.jobBlocks ? L.head : 0u, L.wantJobCost ? c->dJobCost : nullptr, L.jobClk, st2);
        if (L.jobBlocks && le == hipSuccess) le = join(st2);
    } else if (L.jobBlocks) {
        le = launch_table(c->dJobDesc, L.jobBlocks, L.jobBlocksWide, nullptr);
    } else if (L.blockDesc && L.nBlocks) {
        le = launch_table(L.blockDesc, L.nBlocks, L.nBlocksWide, L.wantCost ? c->dTileCost : nullptr);
    } else {
        le = crt_launch_render(&c->hScene, L.slab, c->dCounters, c->dTileClocks, c->dTileOrder, c->tileFirst, c->tileStride, c->tileCount, (uint32_t)c->tilesX,
                               L.sppFirst, L.nf, L.passes, c->ldsBytes, c->cfg.collectStats, nullptr, 0u,
                               L.wantCost ? c->dTileCost : (L.wantJobCost ? c->dJobCost : nullptr), 0u, L.jobClk, st);
    }
    if (L.jobBlocks && le == hipSuccess) c->splitLaunches++;
    return le;
}

// The first single-window launch after a camera / scene change used to run one wavefront per tile (34 ms for the 720p bunny) because nothing was known about the
// tiles.  Now a probe launch (render_narrow_kernel<.., 1>: 512 paths per tile, two through every pixel, one per lane) counts the steps those paths take and stage 0 of the latency mode is a
// block table solved from that estimate (solve_block_table).  The call waits for the probe (the only host wait of crt_render, once per camera / scene).
static int probe_tile_costs(crt_ctx* c, hipStream_t st)
{
    const uint32_t n = c->tileCount;
    if (!c->dTileCost) {
        HIPCK(c, hipMalloc((void**)&c->dTileCost, (size_t)n * 4));
        HIPCK(c, hipHostMalloc((void**)&c->hTileCost, (size_t)n * 4, hipHostMallocDefault));
        HIPCK(c, hipEventCreateWithFlags(&c->costCopied, hipEventDisableTiming));
    } else HIPCK(c, hipEventSynchronize(c->costCopied));                   // an earlier measurement that a camera change abandoned
    const auto tp0 = std::chrono::steady_clock::now();
    HIPCK(c, crt_launch_probe(&c->hScene, c->tileFirst, c->tileStride, n, (uint32_t)c->tilesX, c->dTileCost, st));
    HIPCK(c, hipMemcpyAsync(c->hTileCost, c->dTileCost, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCK(c, hipEventRecord(c->costCopied, st));
    HIPCK(c, hipStreamSynchronize(st));
    // steps of the probe's paths -> an estimate in the units the tuner's model uses: only ratios matter (aims are fractions of the most expensive tile)
    const auto tp1 = std::chrono::steady_clock::now();
    std::vector<uint32_t>& est = c->latCost[0]; est.assign(c->hTileCost, c->hTileCost + n);
    // measured (tools/probe_quality.py; bunny, watch-tower, two-level scene): a tile's one-wavefront duration is AFFINE in the probe's step count — 84 - 113 ticks per step
    // plus 3.7 - 4.8 ms that every tile pays for its 16 384 paths whatever they hit (ray generation, the sky lookup, the sample store): 15 - 22 steps' worth per probed path.
    // With the constant added the estimate ranks the expensive tiles to 6 - 9 % (without: 10 - 20 %, the cheap half of them overrated).
    for (uint32_t i = 0; i < n; i++) est[i] += 18u * crt_probe_paths();
    std::vector<uint8_t>& L = c->latL[0]; L.assign(n, 64);
    if (const char* e = hook("CRT_LAT_PROBE_AIM")) {                       // diagnostics: a fixed fraction of the 98th percentile (the first form of the probed stage)
        uint32_t top = 0;
        { std::vector<uint32_t> sorted(est); const size_t k = (size_t)((double)n * 0.98); std::nth_element(sorted.begin(), sorted.begin() + std::min<size_t>(k, n - 1), sorted.end()); top = sorted[std::min<size_t>(k, n - 1)]; }
        const double aim = atof(e);
        if (top > 0) for (uint32_t i = 0; i < n; i++) L[i] = next_lanes(64, est[i], aim * (double)top);
    } else {
        const std::vector<uint8_t> wide(n, 64);
        solve_block_table(wide, est, lat_budget(c), L);
    }
    c->latProbed = true;
    bool any = false; for (uint32_t i = 0; i < n; i++) any = any || L[i] != 64;
    if (!any) { c->latProbed = false; return 0; }                           // nothing to narrow: stage 0 stays one wavefront per tile
    const auto tp2 = std::chrono::steady_clock::now();
    const int r = upload_block_table(c, L, est);
    if (hook("CRT_LAT_VERBOSE")) { const auto tp3 = std::chrono::steady_clock::now(); auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[crt] cost probe: launch + wait %.3f ms, table solved in %.3f ms, uploaded in %.3f ms\n", ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3)); }
    return r;
}

int crt_render(crt_ctx* c, uint32_t spp_first, uint32_t frames, uint32_t passes)
{
    if (!c) return CRT_ERR_INVALID;
    if (!c->haveScene && !c->havePrim) return c->fail(CRT_ERR_STATE, "crt_render before crt_upload_scene");
    if (passes < 1 || passes > 4) return c->fail(CRT_ERR_INVALID, "passes must be 1..4 (the reference's UI range, renderer.cpp:178)");
    HIPCK(c, hipSetDevice(c->cfg.device));
    if (c->tileCount == 0 || frames == 0) return CRT_OK;
    { int r = update_tile_order(c); if (r) return r; }
    if (c->streams.empty()) {
        int n = c->cfg.renderStreams;
        if (n <= 0) n = 7;
        if (n > 16) n = 16;
        if (c->cfg.collectStats) n = 1;                       // per-tile clocks of a statistics context describe ONE launch
        c->streams.resize((size_t)n);
        for (auto& st : c->streams) HIPCK(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    }
    uint32_t maxF = 0;
    { int r = ensure_pool(c, frames, passes, &maxF); if (r) return r; }
    const size_t windowBytes = window_bytes(c, passes);
    if (c->renderAccel != 0 || c->havePrim) {
        // Renderer::Sample through FileScene's KD-tree / grid, or over the PrimitiveScene (crt_set_render_accel): the sequential form — one wavefront per (tile, window), lane = frame — on the
        // main stream, followed by the ordered accumulate; no planning, no latency mode (the BASELINE configurations are BVH-SAH; this path exists for parity
        // with the reference's shipped FileScene, which traces through its KD-tree: file_scene.h:10-12)
        for (uint32_t f0 = 0; f0 < frames; f0 += maxF) {
            const uint32_t nf = (frames - f0 < maxF) ? frames - f0 : maxF;
            size_t off = 0; int r;
            if ((r = take_region(c, (size_t)((nf + 63u) / 64u) * windowBytes, c->stream, &off))) return r;
            if (c->sceneReady) HIPCK(c, hipStreamWaitEvent(c->stream, c->sceneReady, 0));
            void* slab = c->pool + off;
            if (c->havePrim) HIPCK(c, crt_launch_render_prim(&c->hScene, &c->prim, slab, c->dCounters, c->tileFirst, c->tileStride, c->tileCount, (uint32_t)c->tilesX, spp_first + f0 * passes, nf, passes, c->stream));
            else HIPCK(c, crt_launch_render_alt(c->renderAccel, &c->hScene, &c->alt, slab, c->dCounters, c->tileFirst, c->tileStride, c->tileCount, (uint32_t)c->tilesX, spp_first + f0 * passes, nf, passes, c->stream));
            HIPCK(c, crt_launch_accumulate(slab, c->dAcc, c->tileFirst, c->tileStride, c->tileCount, (uint32_t)c->tilesX, (uint32_t)c->cfg.width, nf, passes, c->stream));
            crt_ctx::Region reg; reg.off = off; reg.bytes = (size_t)((nf + 63u) / 64u) * windowBytes;
            if (c->freeEvents.empty()) HIPCK(c, hipEventCreateWithFlags(&reg.freed, hipEventDisableTiming));
            else { reg.freed = c->freeEvents.back(); c->freeEvents.pop_back(); }
            HIPCK(c, hipEventRecord(reg.freed, c->stream));
            c->inflight.push_back(reg);
        }
        return CRT_OK;
    }
    for (uint32_t f0 = 0; f0 < frames; f0 += maxF) {
        const uint32_t nf = (frames - f0 < maxF) ? frames - f0 : maxF;
        hipStream_t st = c->streams[(size_t)(c->launchSeq++ % c->streams.size())];
        size_t off = 0; int r;
        if ((r = take_region(c, (size_t)((nf + 63u) / 64u) * windowBytes, st, &off))) return r;
        if (c->orderReady) HIPCK(c, hipStreamWaitEvent(st, c->orderReady, 0));
        if (c->sceneReady) HIPCK(c, hipStreamWaitEvent(st, c->sceneReady, 0));
        void* slab = c->pool + off;
        EventPair ev;
        harvest_tuning(c);
        fold_completed(c, c->evRender, &c->foldedRenderMs, &c->foldedLaunches); fold_completed(c, c->evAcc, &c->foldedAccMs, nullptr);
        if ((r = take_event(c, c->evRender, &ev))) return r;
        // the pair is in the timing list from here on; any error exit before its end event is recorded takes it back (a half-recorded pair would make every
        // later crt_get_timing fail in hipEventElapsedTime)
        struct PairGuard { crt_ctx* c; bool armed; ~PairGuard() { if (armed && !c->evRender.empty()) { c->evPool.push_back(c->evRender.back()); c->evRender.pop_back(); } } } pairGuard{c, true};
        Launch L; L.st = st; L.ev = ev; L.slab = slab; L.sppFirst = spp_first + f0 * passes; L.nf = nf; L.passes = passes; L.windows = (nf + 63u) / 64u;
        if ((r = tuner_prepare(c, L))) return r;                       // single-window launches: latency mode (block table, tile-cost measurement)
        if ((r = planner_prepare(c, L))) return r;                     // jobs: tile-cost measurement, kernel choice, plan (block table + pool split) or plain launch
        HIPCK(c, hipEventRecord(ev.a, st));
        const bool pool = L.pool, wantCost = L.wantCost, wantJobCost = L.wantJobCost;
        const hipError_t le = hook("CRT_DEBUG_FAIL_LAUNCH") ? hipErrorInvalidConfiguration /* tests: the runtime refuses the launch */ : launch_render_kernels(c, L);
        if (le != hipSuccess) {
            // a launch that failed has rendered nothing: its timing pair goes back (pairGuard), the accumulator and the region bookkeeping stay untouched —
            // the frames before it are in, this one and the rest are not — and the error is reported
            return c->hip(le, pool ? "launch of render_pool_kernel" : "launch of render_tiles_kernel");
        }
        if (pool) c->poolLaunches++;
        HIPCK(c, hipEventRecord(ev.b, st));
        pairGuard.armed = false;
        c->lastRenderEnd = ev.b;
        if (wantJobCost) {
            HIPCK(c, hipMemcpyAsync(c->hJobCost, c->dJobCost, job_cost_bytes(c), hipMemcpyDeviceToHost, st));
            HIPCK(c, hipEventRecord(c->jobCostCopied, st));
            c->jobCostPending = true;
        }
        if (wantCost && !pool && !c->latDone) {                                                // the tile costs travel to the host behind the launch; looked at by a later crt_render
            HIPCK(c, hipMemcpyAsync(c->hTileCost, c->dTileCost, (size_t)c->tileCount * 4, hipMemcpyDeviceToHost, st));
            HIPCK(c, hipEventRecord(c->costCopied, st));
            c->costPending = true;
        }
        // ordered accumulation on the main stream (frame order = launch order), behind this launch
        HIPCK(c, hipStreamWaitEvent(c->stream, ev.b, 0));
        if ((r = take_event(c, c->evAcc, &ev))) return r;
        HIPCK(c, hipEventRecord(ev.a, c->stream));
        HIPCK(c, crt_launch_accumulate(slab, c->dAcc, c->tileFirst, c->tileStride, c->tileCount, (uint32_t)c->tilesX, (uint32_t)c->cfg.width, nf, passes, c->stream));
        HIPCK(c, hipEventRecord(ev.b, c->stream));
        crt_ctx::Region reg; reg.off = off; reg.bytes = (size_t)((nf + 63u) / 64u) * windowBytes;
        if (c->freeEvents.empty()) HIPCK(c, hipEventCreateWithFlags(&reg.freed, hipEventDisableTiming));
        else { reg.freed = c->freeEvents.back(); c->freeEvents.pop_back(); }
        HIPCK(c, hipEventRecord(reg.freed, c->stream));
        c->inflight.push_back(reg);
    }
    return CRT_OK;
}

int crt_sync(crt_ctx* c)
{
    if (!c) return CRT_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipStreamSynchronize(c->stream));          // every render launch is followed by its accumulate on this stream
    for (auto st : c->streams) HIPCK(c, hipStreamSynchronize(st));
    return CRT_OK;
}

int crt_clear(crt_ctx* c)
{
    if (!c) return CRT_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipMemsetAsync(c->dAcc, 0, (size_t)c->cfg.width * c->cfg.height * 16, c->stream));
    return CRT_OK;
}

int crt_read_accumulator(crt_ctx* c, float* host)
{
    if (!c || !host) return CRT_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipMemcpyAsync(host, c->dAcc, (size_t)c->cfg.width * c->cfg.height * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return CRT_OK;
}

int crt_resolve_screen(crt_ctx* c, float scale, uint32_t* hostPixels, float* energy)
{
    if (!c) return CRT_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->cfg.device));
    const int tiles = c->tilesX * c->tilesY;
    HIPCK(c, crt_launch_resolve(c->dAcc, c->dPixels, c->dTileSums, c->tileFirst, c->tileStride, c->tileCount, (uint32_t)c->tilesX, (uint32_t)c->cfg.width, scale, c->stream));
    if (hostPixels) HIPCK(c, hipMemcpyAsync(hostPixels, c->dPixels, (size_t)c->cfg.width * c->cfg.height * 4, hipMemcpyDeviceToHost, c->stream));
    std::vector<float> sums(tiles);
    HIPCK(c, hipMemcpyAsync(sums.data(), c->dTileSums, (size_t)tiles * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (energy) { float e = 0; for (int i = 0; i < tiles; i++) e += sums[i]; *energy = e; }   // renderer.cpp:155-157, tile order
    return CRT_OK;
}

int crt_whitted_tick(crt_ctx* c, uint32_t* hostPixels)
{
    if (!c) return CRT_ERR_INVALID;
    if (!c->haveScene) return c->fail(CRT_ERR_STATE, "crt_whitted_tick before crt_upload_scene");
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, crt_launch_whitted(&c->hScene, c->renderAccel, &c->alt, c->dAcc, c->dPixels, c->dCounters, c->ldsBytes, c->stream));
    if (hostPixels) HIPCK(c, hipMemcpyAsync(hostPixels, c->dPixels, (size_t)c->cfg.width * c->cfg.height * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return CRT_OK;
}

int crt_find_nearest(crt_ctx* c, const crt_ray* rays, crt_hit* hits, size_t n)
{
    if (!c || (n && (!rays || !hits))) return CRT_ERR_INVALID;
    if (!c->haveScene && !c->havePrim) return c->fail(CRT_ERR_STATE, "crt_find_nearest before crt_upload_scene");
    if (n == 0) return CRT_OK;
    if (n > 0x7fffffffull) return c->fail(CRT_ERR_UNSUPPORTED, "at most 2^31-1 rays per call");
    HIPCK(c, hipSetDevice(c->cfg.device));
    // query buffers are kept and grown to the high-water mark (hipMalloc synchronises the device)
    if (n > c->queryCap) {
        HIPCK(c, hipStreamSynchronize(c->stream));
        if (c->dQueryRays) (void)hipFree(c->dQueryRays);
        if (c->dQueryHits) (void)hipFree(c->dQueryHits);
        c->dQueryRays = c->dQueryHits = nullptr; c->queryCap = 0;
        HIPCK(c, hipMalloc(&c->dQueryRays, n * sizeof(crt_ray)));
        HIPCK(c, hipMalloc(&c->dQueryHits, n * sizeof(crt_hit)));
        c->queryCap = n;
    }
    void *dR = c->dQueryRays, *dH = c->dQueryHits;
    HIPCK(c, hipMemcpyAsync(dR, rays, n * sizeof(crt_ray), hipMemcpyHostToDevice, c->stream));
    if (c->havePrim) HIPCK(c, crt_launch_find_nearest_prim(&c->prim, dR, dH, (uint32_t)n, c->stream));
    else HIPCK(c, crt_launch_find_nearest(&c->hScene, dR, dH, (uint32_t)n, c->dCounters, c->ldsBytes, c->dQueryCursor, c->stream));
    HIPCK(c, hipMemcpyAsync(hits, dH, n * sizeof(crt_hit), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return CRT_OK;
}

// FileScene's KD-tree / uniform grid over the scene's triangles, for crt_find_nearest_alt
int crt_upload_alt_accel(crt_ctx* c, const crt_alt_accel* a)
{
    if (!c || !a) return CRT_ERR_INVALID;
    if (!c->haveScene || c->hScene.kind != CRT_SCENE_FILE) return c->fail(CRT_ERR_STATE, "crt_upload_alt_accel needs an uploaded CRT_SCENE_FILE scene (light quad, floor plane, materials)");
    if (a->kind != CRT_ACCEL_KDTREE && a->kind != CRT_ACCEL_GRID) return c->fail(CRT_ERR_INVALID, "unknown accelerator kind %d", a->kind);
    if (!a->triangles || a->triCount == 0) return c->fail(CRT_ERR_INVALID, "accelerator has no triangles");
    HIPCK(c, hipSetDevice(c->cfg.device));
    const int slot = a->kind == CRT_ACCEL_KDTREE ? 0 : 1;
    uint32_t kdHeight = 0;
    if (a->kind == CRT_ACCEL_KDTREE) {
        if (!a->kdNodes || a->kdNodeCount == 0 || (a->kdTriIndexCount && !a->kdTriIndices)) return c->fail(CRT_ERR_INVALID, "KD-tree arrays missing");
        std::vector<std::pair<uint32_t, uint32_t>> st; st.push_back({0u, 0u}); size_t visited = 0;
        while (!st.empty()) {
            auto [n, d] = st.back(); st.pop_back();
            if (++visited > (size_t)a->kdNodeCount) return c->fail(CRT_ERR_INVALID, "KD node graph is not a tree");
            const crt_kd_node& nd = a->kdNodes[n];
            if (d > kdHeight) kdHeight = d;
            if (nd.left < 0) { if ((uint64_t)nd.firstTri + nd.triCount > a->kdTriIndexCount) return c->fail(CRT_ERR_INVALID, "KD leaf %u: triangle range out of bounds", n); continue; }
            if (nd.right < 0 || (uint32_t)nd.left >= a->kdNodeCount || (uint32_t)nd.right >= a->kdNodeCount || nd.splitAxis < 0 || nd.splitAxis > 2)
                return c->fail(CRT_ERR_INVALID, "KD node %u: child index / split axis out of range", n);
            st.push_back({(uint32_t)nd.left, d + 1}); st.push_back({(uint32_t)nd.right, d + 1});
        }
        for (uint32_t i = 0; i < a->kdTriIndexCount; i++) if (a->kdTriIndices[i] >= a->triCount) return c->fail(CRT_ERR_INVALID, "kdTriIndices[%u] out of range", i);
        if ((kdHeight + 1) * 128u * 4u > 64u * 1024u) return c->fail(CRT_ERR_UNSUPPORTED, "KD-tree height %u exceeds the LDS traversal stack", kdHeight);
    } else {
        uint64_t cells = 1;
        for (int k = 0; k < 3; k++) { if (a->gridResolution[k] < 1 || a->gridResolution[k] > 128) return c->fail(CRT_ERR_INVALID, "grid resolution must be 1..128 per axis (grid.cpp:22)"); cells *= (uint64_t)a->gridResolution[k]; }
        if (!a->gridCellStart || (a->gridCellTriCount && !a->gridCellTris)) return c->fail(CRT_ERR_INVALID, "grid arrays missing");
        if (a->gridCellStart[0] != 0 || a->gridCellStart[cells] != a->gridCellTriCount) return c->fail(CRT_ERR_INVALID, "gridCellStart must run from 0 to gridCellTriCount");
        for (uint64_t i = 0; i < cells; i++) if (a->gridCellStart[i] > a->gridCellStart[i + 1]) return c->fail(CRT_ERR_INVALID, "gridCellStart is not monotone at cell %llu", (unsigned long long)i);
        for (uint32_t i = 0; i < a->gridCellTriCount; i++) if (a->gridCellTris[i] < 0 || (uint32_t)a->gridCellTris[i] >= a->triCount) return c->fail(CRT_ERR_INVALID, "gridCellTris[%u] out of range", i);
    }
    HIPCK(c, hipStreamSynchronize(c->stream));                            // queries of the previous structure
    for (void* p : c->altAllocs[slot]) (void)hipFree(p);
    c->altAllocs[slot].clear();
    if (slot == 0) c->haveKd = false; else c->haveGrid = false;
    if (c->renderAccel == a->kind) c->renderAccel = 0;
    auto up = [&](const void* src, size_t bytes, const void** out) -> int {
        *out = nullptr; if (!bytes) return 0;
        void* d = nullptr; HIPCK(c, hipMalloc(&d, bytes)); c->altAllocs[slot].push_back(d);
        HIPCK(c, hipMemcpy(d, src, bytes, hipMemcpyHostToDevice)); *out = d; return 0;
    };
    int r;
    // Möller–Trumbore operands in the reference's triangle order (one array shared by both structures)
    if (!c->altTris || c->altTriCount != a->triCount) {
        if (c->altTris) { (void)hipFree(c->altTris); c->altTris = nullptr; }
        HIPCK(c, hipMalloc(&c->altTris, (size_t)a->triCount * 48)); c->altTriCount = a->triCount;
    }
    {
        std::vector<float> rec((size_t)a->triCount * 12);
        for (uint32_t i = 0; i < a->triCount; i++) {
            const crt_tri& t = a->triangles[i]; float* o = &rec[(size_t)i * 12];
            for (int k = 0; k < 3; k++) { o[k] = t.vertex0[k]; o[4 + k] = t.vertex1[k] - t.vertex0[k]; o[8 + k] = t.vertex2[k] - t.vertex0[k]; }
            memcpy(&o[3], &i, 4); memcpy(&o[7], &t.objIdx, 4); o[11] = 0;
        }
        HIPCK(c, hipMemcpy(c->altTris, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
        c->alt.tris = c->altTris;
    }
    if (slot == 0) {
        if ((r = up(a->kdNodes, (size_t)a->kdNodeCount * 48, &c->alt.kdNodes))) return r;
        const void* p = nullptr; if ((r = up(a->kdTriIndices, (size_t)a->kdTriIndexCount * 4, &p))) return r;
        c->alt.kdRefs = (const uint32_t*)p; c->alt.kdStack = kdHeight + 1; c->haveKd = true;
    } else {
        uint64_t cells = (uint64_t)a->gridResolution[0] * a->gridResolution[1] * a->gridResolution[2];
        const void* p = nullptr;
        if ((r = up(a->gridCellStart, (size_t)(cells + 1) * 4, &p))) return r; c->alt.cellStart = (const uint32_t*)p;
        if ((r = up(a->gridCellTris, (size_t)a->gridCellTriCount * 4, &p))) return r; c->alt.cellRefs = (const int32_t*)p;
        for (int k = 0; k < 3; k++) { c->alt.res[k] = a->gridResolution[k]; c->alt.cell[k] = a->gridCellSize[k]; c->alt.lo[k] = a->gridMin[k]; c->alt.hi[k] = a->gridMax[k]; }
        c->haveGrid = true;
    }
    return CRT_OK;
}

int crt_upload_primitive_scene(crt_ctx* c, const crt_primitive_scene* ps)
{
    if (!c || !ps) return CRT_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->cfg.device));
    for (const crt_texture* t : {&ps->red, &ps->blue})
        if (t->pixels && (t->width != 512 || t->height != 512)) return c->fail(CRT_ERR_INVALID, "PrimitiveScene wall images are 512 x 512 (Plane::GetAlbedo masks the texel coordinates with 511)");
    for (auto st : c->streams) HIPCK(c, hipStreamSynchronize(st));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->freeScene();
    crt::PrimDev& p = c->prim; p = crt::PrimDev{};
    memcpy(p.quadInvT, ps->quadInvT, 48); p.quadNrm[0] = -ps->quadT[1]; p.quadNrm[1] = -ps->quadT[5]; p.quadNrm[2] = -ps->quadT[9]; p.quadSize = ps->quadSize;   // Quad::GetNormal, primitives.h:363-367
    memcpy(p.spherePos, ps->spherePos, 12);
    memcpy(p.cubeInvM, ps->cubeInvM, 48); memcpy(p.cubeM, ps->cubeM, 48); memcpy(p.cubeMin, ps->cubeMin, 12); memcpy(p.cubeMax, ps->cubeMax, 12);
    memcpy(p.torusInvT, ps->torusInvT, 48); memcpy(p.torusT, ps->torusT, 48); p.rt2 = ps->torusRt2; p.rc2 = ps->torusRc2; p.r2 = ps->torusR2;
    memcpy(p.refl, ps->reflectivity, 44); memcpy(p.refr, ps->refractivity, 44); memcpy(p.absorb, ps->absorption, 132);
    HIPCK(c, hipMalloc((void**)&c->dPrimTex, 2u * 512u * 512u * 4u));
    HIPCK(c, hipMemsetAsync(c->dPrimTex, 0, 2u * 512u * 512u * 4u, c->stream));
    if (ps->red.pixels) HIPCK(c, hipMemcpyAsync(c->dPrimTex, ps->red.pixels, 512u * 512u * 4u, hipMemcpyHostToDevice, c->stream));
    if (ps->blue.pixels) HIPCK(c, hipMemcpyAsync(c->dPrimTex + 512u * 512u, ps->blue.pixels, 512u * 512u * 4u, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    p.red = c->dPrimTex; p.blue = c->dPrimTex + 512u * 512u;
    c->havePrim = true; c->orderDirty = false;
    return CRT_OK;
}

int crt_set_render_accel(crt_ctx* c, int kind)
{
    if (!c) return CRT_ERR_INVALID;
    if (kind != 0 && kind != CRT_ACCEL_KDTREE && kind != CRT_ACCEL_GRID) return c->fail(CRT_ERR_INVALID, "crt_set_render_accel: unknown accelerator kind %d", kind);
    if ((kind == CRT_ACCEL_KDTREE && !c->haveKd) || (kind == CRT_ACCEL_GRID && !c->haveGrid)) return c->fail(CRT_ERR_STATE, "crt_set_render_accel: no such accelerator uploaded (kind %d)", kind);
    if (kind != 0 && (c->alt.kdStack * 2u + 15u) * 64u * 4u * 4u > 64u * 1024u) return c->fail(CRT_ERR_UNSUPPORTED, "KD-tree height %u exceeds the render kernel's LDS stack", c->alt.kdStack);
    c->renderAccel = kind;
    return CRT_OK;
}

int crt_find_nearest_alt(crt_ctx* c, int kind, const crt_ray* rays, crt_hit* hits, size_t n)
{
    if (!c || (n && (!rays || !hits))) return CRT_ERR_INVALID;
    if (!((kind == CRT_ACCEL_KDTREE && c->haveKd) || (kind == CRT_ACCEL_GRID && c->haveGrid))) return c->fail(CRT_ERR_STATE, "crt_find_nearest_alt: no such accelerator uploaded (kind %d)", kind);
    if (n == 0) return CRT_OK;
    if (n > 0x7fffffffull) return c->fail(CRT_ERR_UNSUPPORTED, "at most 2^31-1 rays per call");
    HIPCK(c, hipSetDevice(c->cfg.device));
    if (n > c->queryCap) {
        HIPCK(c, hipStreamSynchronize(c->stream));
        if (c->dQueryRays) (void)hipFree(c->dQueryRays);
        if (c->dQueryHits) (void)hipFree(c->dQueryHits);
        c->dQueryRays = c->dQueryHits = nullptr; c->queryCap = 0;
        HIPCK(c, hipMalloc(&c->dQueryRays, n * sizeof(crt_ray)));
        HIPCK(c, hipMalloc(&c->dQueryHits, n * sizeof(crt_hit)));
        c->queryCap = n;
    }
    HIPCK(c, hipMemcpyAsync(c->dQueryRays, rays, n * sizeof(crt_ray), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, crt_launch_find_nearest_alt(kind, &c->hScene, &c->alt, c->dQueryRays, c->dQueryHits, (uint32_t)n, c->dQueryCursor, c->stream));
    HIPCK(c, hipMemcpyAsync(hits, c->dQueryHits, n * sizeof(crt_hit), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return CRT_OK;
}

int crt_get_counters(crt_ctx* c, crt_counters* out)
{
    if (!c || !out) return CRT_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipMemcpyAsync(out, c->dCounters, sizeof(crt::Counters), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return CRT_OK;
}

int crt_reset_counters(crt_ctx* c)
{
    if (!c) return CRT_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipMemsetAsync(c->dCounters, 0, sizeof(crt::Counters), c->stream));
    return CRT_OK;
}

int crt_get_timing(crt_ctx* c, crt_timing* out)
{
    if (!c || !out) return CRT_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipStreamSynchronize(c->stream));
    for (auto st : c->streams) HIPCK(c, hipStreamSynchronize(st));
    harvest_tuning(c);                                         // the latency mode's stage timings, before the pairs are recycled
    memset(out, 0, sizeof(*out));
    out->render_kernel_ms = (float)c->foldedRenderMs; out->resolve_kernel_ms = (float)c->foldedAccMs;
    for (auto& ev : c->evRender) { float ms = 0; HIPCK(c, hipEventElapsedTime(&ms, ev.a, ev.b)); out->render_kernel_ms += ms; }
    for (auto& ev : c->evAcc) { float ms = 0; HIPCK(c, hipEventElapsedTime(&ms, ev.a, ev.b)); out->resolve_kernel_ms += ms; }
    out->render_launches = (uint32_t)c->evRender.size() + c->foldedLaunches;
    out->pool_launches = c->poolLaunches; out->split_launches = c->splitLaunches;
    c->foldedRenderMs = c->foldedAccMs = 0; c->foldedLaunches = 0; c->poolLaunches = 0; c->splitLaunches = 0;
    for (auto& ev : c->evRender) c->evPool.push_back(ev);      // the figures cover every launch since the previous crt_get_timing
    for (auto& ev : c->evAcc) c->evPool.push_back(ev);
    c->evRender.clear(); c->evAcc.clear();
    return CRT_OK;
}

int crt_get_tile_clocks(crt_ctx* c, uint64_t* out)
{
    if (!c || !out) return CRT_ERR_INVALID;
    if (!c->dTileClocks) return c->fail(CRT_ERR_STATE, "tile clocks are recorded only by a collectStats context");
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipMemcpyAsync(out, c->dTileClocks, (size_t)c->tileCount * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return CRT_OK;
}

// diagnostic builds (-DCRT_STAMPS) only: 16 extra words per tile behind the tile clocks (not part of the public ABI)
// tests: HIP timing events currently held by the context (bounded: completed launches are folded into totals, see fold_completed)
extern "C" int crt_debug_live_events(crt_ctx* c) { return c ? (int)(2 * (c->evRender.size() + c->evAcc.size() + c->evPool.size())) : -1; }

// tests: the device's short reciprocals (dev_common.h rcp_exact*) against the IEEE division over all 2^32 inputs; out[4] = {inputs, differences x 3}
extern "C" int crt_debug_check_reciprocals(crt_ctx* c, uint64_t* out)
{
    if (!c || !out) return CRT_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->cfg.device));
    unsigned long long* d = nullptr;
    HIPCK(c, hipMalloc(&d, 32));
    hipError_t e = hipMemsetAsync(d, 0, 32, c->stream);
    if (e == hipSuccess) e = crt_launch_check_reciprocals(d, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, 32, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIPCK(c, e);
    return CRT_OK;
}

// tests (no GPU needed): the planner's decision for given tile costs — plan_job on a context that holds nothing but the costs.  table[] receives the block
// descriptors (tile | first frame << 16 | log2(lanes) << 22 | window << 25), order[] the dispatch order, *head the number of leading tiles of it that go to the table
// when the rest goes to the pool.  Returns the number of blocks, or a negative error code.
extern "C" int crt_debug_plan_job(const uint32_t* cost, uint32_t n, uint32_t windows, uint32_t frames, int pool, double poolWindowTicks,
                                  uint32_t* table, uint32_t tableCap, uint32_t* head, uint32_t* order)
{
    if (!cost || !n || !table || !head || !order) return CRT_ERR_INVALID;
    crt_ctx c;
    c.tileCount = n; c.jobCostValid = true; c.poolWindowTicks = poolWindowTicks;
    c.streams.assign(2, nullptr);
    c.jobCost.assign(cost, cost + n);
    c.jobOrder.resize(n);
    for (uint32_t i = 0; i < n; i++) c.jobOrder[i] = i;
    std::stable_sort(c.jobOrder.begin(), c.jobOrder.end(), [&](uint32_t a, uint32_t b) { return c.jobCost[a] > c.jobCost[b]; });
    std::vector<uint32_t> t; uint32_t h = 0;
    plan_job(&c, windows, frames, pool != 0, t, &h);
    c.streams.clear();
    if (t.size() > tableCap) return CRT_ERR_INVALID;
    memcpy(table, t.data(), t.size() * 4); memcpy(order, c.jobOrder.data(), (size_t)n * 4); *head = h;
    return (int)t.size();
}

// diagnostics (tools/latency_probe.py): the tile costs the last recording single-window launch left on the device (100 MHz ticks, longest wavefront per tile)
extern "C" int crt_debug_tile_costs(crt_ctx* c, uint32_t* out)
{
    if (!c || !out || !c->dTileCost) return CRT_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipDeviceSynchronize());
    HIPCK(c, hipMemcpy(out, c->dTileCost, (size_t)c->tileCount * 4, hipMemcpyDeviceToHost));
    return CRT_OK;
}

// diagnostics (tools/probe_quality.py): lanes per wavefront and tile costs of one stage of the latency tuner ([0]: the probe's estimates when latProbed); returns the installed stage
extern "C" int crt_debug_lat_stage(crt_ctx* c, int stage, uint8_t* lanes, uint32_t* cost)
{
    if (!c || stage < 0 || stage > crt_ctx::kLatStages) return CRT_ERR_INVALID;
    if (c->latL[stage].size() != c->tileCount || c->latCost[stage].size() != c->tileCount) return CRT_ERR_STATE;
    if (lanes) memcpy(lanes, c->latL[stage].data(), c->tileCount);
    if (cost) memcpy(cost, c->latCost[stage].data(), (size_t)c->tileCount * 4);
    return c->latStage;
}

extern "C" int crt_debug_tile_stamps(crt_ctx* c, uint64_t* out)
{
    if (!c || !out || !c->dTileClocks) return CRT_ERR_INVALID;
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipMemcpyAsync(out, c->dTileClocks + 2 * (size_t)c->tileCount, (size_t)c->tileCount * 128, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return CRT_OK;
}

int crt_bind_accumulator(crt_ctx* c, void* p)
{
    if (!c) return CRT_ERR_INVALID;
    if (p && ((uintptr_t)p & 15u)) return c->fail(CRT_ERR_INVALID, "accumulator pointer must be 16-byte aligned");
    HIPCK(c, hipSetDevice(c->cfg.device));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->dAcc = p ? p : c->dAccOwned;
    return CRT_OK;
}

int crt_accumulator_device_ptr(crt_ctx* c, void** p)
{
    if (!c || !p) return CRT_ERR_INVALID;
    *p = c->dAcc; return CRT_OK;
}

} // extern "C"
