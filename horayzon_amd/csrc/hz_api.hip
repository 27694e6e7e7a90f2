// hz_api.hip -- C-ABI entry points of libhorayzon_hip.so (see include/horayzon_hip.h).
//
// Host-side driver logic restating horizon_gridded_comp (horizon_comp.cpp:629-822)
// and CppTerrain (shadow_comp.cpp:304-605): unit conversions, trig tables with the
// reference's float/double promotion pattern, uploads, kernel launches, reports.
#include "hz_internal.h"
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <limits>
#include <mutex>
#include <utility>
#include <thread>
#include <atomic>

namespace hz {

static thread_local std::string g_error;

int set_error(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    const hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// device view of an input array that may live on the host
template <typename T>
struct DevIn {
    const T *dev = nullptr;
    void *owned = nullptr;
    ~DevIn() { if (owned) (void)hipFree(owned); }
    int bind(const T *src, size_t count, hipStream_t st) {
        if (!src || count == 0) { dev = nullptr; return HZ_OK; }
        if (is_device_ptr(src)) { dev = src; return HZ_OK; }
        HZ_HIP(hipMalloc(&owned, count * sizeof(T)));
        HZ_HIP(hipMemcpyAsync(owned, src, count * sizeof(T), hipMemcpyHostToDevice, st));
        dev = static_cast<const T *>(owned);
        return HZ_OK;
    }
};

// device buffer for an output array that may live on the host
template <typename T>
struct DevOut {
    T *dev = nullptr;
    T *host = nullptr;
    void *owned = nullptr;
    size_t count = 0;
    ~DevOut() { if (owned) (void)hipFree(owned); }
    int bind(T *dst, size_t n) {
        count = n;
        if (!dst || n == 0) { dev = nullptr; return HZ_OK; }
        if (is_device_ptr(dst)) { dev = dst; return HZ_OK; }
        HZ_HIP(hipMalloc(&owned, n * sizeof(T)));
        dev = static_cast<T *>(owned); host = dst;
        return HZ_OK;
    }
    int finish(hipStream_t st) {
        if (host && count) HZ_HIP(hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, st));
        return HZ_OK;
    }
};

static int select_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return set_error(HZ_ERR_NODEV, "no HIP device available (libhorayzon_hip needs an MI355X / gfx950 GPU)");
    }
    if (device < 0 || device >= n) return set_error(HZ_ERR_ARG, "device %d out of range (%d devices)", device, n);
    HZ_HIP(hipSetDevice(device));
    return HZ_OK;
}

// ---- helpers restating horizon_comp.cpp:37-39 ------------------------------------------
static inline float deg2rad_f(float ang) { return (float)(((double)ang / 180.0) * M_PI); }

struct HostTables {
    std::vector<float> azim_sin, azim_cos, elev_ang, elev_sin, elev_cos;
    std::vector<int> mid_idx;   // (index nearest to the midpoint of entries i and i + 10, bits of elev_ang[that index]) (horizon_comp.cpp:462-464)
    int elev_num = 0;
    float hori_acc = 0, low = 0, up = 0;
};

// horizon_comp.cpp:648, :667-669, :711-731
static void build_tables(int azim_num, float hori_acc_deg, float low_deg, HostTables &t) {
    float elev_ang_up_lim = 89.98;
    t.hori_acc = deg2rad_f(hori_acc_deg);
    t.low = deg2rad_f(low_deg);
    t.up = deg2rad_f(elev_ang_up_lim);
    t.azim_sin.resize((size_t)azim_num); t.azim_cos.resize((size_t)azim_num);
    float ang;
    for (int i = 0; i < azim_num; i++) {
        ang = (float)(((2 * M_PI) / azim_num) * i);
        t.azim_sin[(size_t)i] = sinf(ang);
        t.azim_cos[(size_t)i] = cosf(ang);
    }
    const double step = (double)t.hori_acc / 5.0;
    t.elev_num = (int)ceil((double)(t.up - t.low) / step) + 1;
    const size_t n = (size_t)std::max(t.elev_num, 0);
    t.elev_ang.resize(n); t.elev_sin.resize(n); t.elev_cos.resize(n);
    for (int i = 0; i < t.elev_num; i++) {
        ang = (float)((double)t.up - step * (double)i);
        t.elev_ang[(size_t)(t.elev_num - i - 1)] = ang;
        t.elev_sin[(size_t)(t.elev_num - i - 1)] = sinf(ang);
        t.elev_cos[(size_t)(t.elev_num - i - 1)] = cosf(ang);
    }
    // elev_samp = (elev_ang[prev] + elev_ang[ind]) / 2.0; ind = (int)roundf((elev_samp - low) / (hori_acc / 5.0))
    // (stored as pairs: the index and the bits of elev_ang[index], the value the search emits -- one 8 B load instead of
    //  two dependent 4 B loads at the end of every azimuth)
    t.mid_idx.assign(2 * (n > 0 ? n : 1), 0);
    for (int i = 0; i + 10 < t.elev_num; i++) {
        const float es = (float)((double)(t.elev_ang[(size_t)i] + t.elev_ang[(size_t)i + 10]) / 2.0);
        const int mid = (int)roundf((float)((double)(es - t.low) / step));
        t.mid_idx[2 * (size_t)i] = mid;
        const float ev = t.elev_ang[(size_t)std::min(std::max(mid, 0), t.elev_num - 1)];
        memcpy(&t.mid_idx[2 * (size_t)i + 1], &ev, sizeof(float));
    }
}

static int parse_alg(const char *s, int *alg) {
    if (s && strcmp(s, "discrete_sampling") == 0) { *alg = 0; return HZ_OK; }
    if (s && strcmp(s, "binary_search") == 0) { *alg = 1; return HZ_OK; }
    if (s && strcmp(s, "guess_constant") == 0) { *alg = 2; return HZ_OK; }
    return set_error(HZ_ERR_ARG, "invalid input argument for ray_algorithm");
}

static int check_geom(const char *s) {
    // all three describe the same surface (horizon_comp.cpp:139-183); the explicit
    // "triangle" split is what the LBVH leaves hold
    if (s && (strcmp(s, "triangle") == 0 || strcmp(s, "quad") == 0 || strcmp(s, "grid") == 0)) return HZ_OK;
    return set_error(HZ_ERR_ARG, "invalid input argument for geom_type");
}

// ---- stream pool --------------------------------------------------------------------------------
// Streams are taken from a per-device pool and go back to it; the library NEVER calls hipStreamDestroy.
// Reason (root cause of the "stray element" of round 1, DESIGN_HISTORY.md section 2): with the HIP runtime of
// ROCm 7.0 (libamdhip64 as bundled with PyTorch 2.10+rocm7.0) hipStreamDestroy() frees the ~920-byte stream
// object while a completion callback of that stream can still be pending on the ROCr async-events thread;
// the callback then decrements a counter at offset 152 and stores a 32-bit zero at offset 888 of the FREED
// block -- i.e. into whatever the application allocated there next (a 912-byte NumPy array shares the malloc
// size class: one float of a result array turned 0.0 after the call had returned).  Caught with the heap
// tripwire scripts/stray/hzq_preload.c: writer = libhsa-runtime64 AsyncEventsLoop -> libamdhip64 callback,
// block allocated in hipStreamCreateWithFlags, freed in hipStreamDestroy.  A pooled stream is synchronised
// before it is reused, so no work ever outlives its owner.
static std::mutex g_stream_mu;
static std::vector<std::pair<int, hipStream_t>> g_stream_pool;    // (device, idle stream)

static int stream_acquire(int device, hipStream_t *st) {
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        for (size_t i = 0; i < g_stream_pool.size(); i++)
            if (g_stream_pool[i].first == device) {
                *st = g_stream_pool[i].second;
                g_stream_pool.erase(g_stream_pool.begin() + (long)i);
                return HZ_OK;
            }
    }
    const hipError_t e = hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    if (e != hipSuccess) return set_error(HZ_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    return HZ_OK;
}

static void stream_release(int device, hipStream_t st) {
    if (!st) return;
    (void)hipStreamSynchronize(st);
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_stream_pool.emplace_back(device, st);
}

static int scene_new(int device, Scene **out) {
    int rc = select_device(device);
    if (rc) return rc;
    Scene *sc = new Scene();
    sc->device = device;
    if ((rc = stream_acquire(device, &sc->stream))) { delete sc; return rc; }
    *out = sc;
    return HZ_OK;
}

static void scene_free(Scene *sc) {
    if (!sc) return;
    (void)hipSetDevice(sc->device);
    stream_release(sc->device, sc->stream);
    if (sc->near_buf) (void)hipFree(sc->near_buf);
    if (sc->left_buf) (void)hipFree(sc->left_buf);
    if (sc->owns_blob && sc->blob) (void)hipFree(sc->blob);
    delete sc;
}

struct Terrain {
    int device = 0;
    Scene *scene = nullptr;
    bool owns_scene = false;
    int offset_0 = 0, offset_1 = 0, dim_in_0 = 0, dim_in_1 = 0;
    void *tilt = nullptr, *norm = nullptr, *enl = nullptr, *elev = nullptr, *mask = nullptr;
    double *refrac_fac = nullptr;             // refrac_cor: per cell the pressure / temperature factor of the refraction formula (hz_shadow.hip)
    bool own_tilt = false, own_norm = false, own_enl = false, own_elev = false, own_mask = false;
    float fill = 0, ang_max = 89.0f;
    int refrac = 0;
    int count_work = 0;                       // hz_terrain_count_work
    unsigned long long *counters = nullptr;   // device u64[16]
    float *sun_dev = nullptr;                 // persistent device copy of the sun positions of a call (grown on demand: the
    size_t sun_cap = 0;                       //   drop-in API is called once per time step -- no hipMalloc / hipFree per call)
    hipStream_t stream = nullptr;
    bool initialised = false;
};

static void terrain_release_arrays(Terrain *t) {
    (void)hipSetDevice(t->device);
    if (t->own_tilt && t->tilt) (void)hipFree(t->tilt);
    if (t->own_norm && t->norm) (void)hipFree(t->norm);
    if (t->own_enl && t->enl) (void)hipFree(t->enl);
    if (t->own_elev && t->elev) (void)hipFree(t->elev);
    if (t->own_mask && t->mask) (void)hipFree(t->mask);
    if (t->refrac_fac) { (void)hipFree(t->refrac_fac); t->refrac_fac = nullptr; }
    t->tilt = t->norm = t->enl = t->elev = t->mask = nullptr;
    t->own_tilt = t->own_norm = t->own_enl = t->own_elev = t->own_mask = false;
    // the counters live on the terrain's current GPU; a re-initialisation may move the terrain to another one
    if (t->counters) { (void)hipFree(t->counters); t->counters = nullptr; }
    if (t->sun_dev) { (void)hipFree(t->sun_dev); t->sun_dev = nullptr; t->sun_cap = 0; }
    if (t->owns_scene && t->scene) scene_free(t->scene);
    t->scene = nullptr; t->owns_scene = false; t->initialised = false;
}

// persistent device copy (Terrain keeps its inputs in HBM instead of raw host pointers,
// shadow_comp.cpp:332-346)
static int persist(const void *src, size_t bytes, hipStream_t st, void **dst, bool *own) {
    if (is_device_ptr(src)) { *dst = const_cast<void *>(src); *own = false; return HZ_OK; }
    HZ_HIP(hipMalloc(dst, bytes ? bytes : 16));
    *own = true;
    HZ_HIP(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st));
    return HZ_OK;
}

// Host output of the drop-in call (NumPy memory, pageable): a device-to-host copy into pageable memory is staged by the
// runtime (copy kernels on the CUs, host-blocking, 16 GB/s into untouched pages -- scripts/d2h_probe.py) and slows the
// traversal kernel it is meant to hide behind.  HostPinner page-locks the caller's slab region by region (one region per
// chunk of rows) on a helper thread, ahead of the copies: 45 ms per GB, done while the first chunks are traced; the copies
// are then plain DMA (57 GB/s, no CU time).
// Only pages that lie WHOLLY inside the slab are ever registered (the first region starts at the first page boundary at or
// after the slab's first byte, the last one ends at the last boundary at or before its end): a neighbouring slab of the
// same array -- horizon.py's devices=[...] path runs one call per GPU on adjacent slabs of one NumPy array -- never sees one
// of its pages registered or unregistered by somebody else.  The unaligned head and tail (< 4096 B each) go as pageable
// copies of their own.  Regions are page aligned and disjoint; a copy is cut at every region boundary it contains (a copy
// must not straddle two separately registered ranges, nor a registered and a pageable one) and waits for every region it
// touches.  If page locking fails (memlock limit, memory that is already registered) the remaining regions stay pageable.
// Outputs below 8 MiB are not pinned at all: the thread and the registration cost more than they save.
struct HostPinner {
    std::vector<std::pair<char *, size_t>> regions;      // what is (still) registered: the worker zeroes a size on failure
    std::vector<std::pair<char *, size_t>> planned;      // the same list as laid out by start()
    std::atomic<int> done{0};
    std::atomic<bool> failed{false};
    std::thread worker;
    int device = 0;
    void start(int dev, char *base, const std::vector<size_t> &chunk_bytes) {
        device = dev;
        const uintptr_t page = 4096;
        size_t total = 0;
        for (size_t b : chunk_bytes) total += b;
        if (total < ((size_t)8 << 20)) return;
        const uintptr_t first = reinterpret_cast<uintptr_t>(base);
        const uintptr_t last_page = (first + total) & ~(page - 1);          // end of the last whole page inside the slab
        uintptr_t lo = (first + page - 1) & ~(page - 1);                    // first whole page inside the slab
        uintptr_t cur = first;
        for (size_t k = 0; k < chunk_bytes.size(); k++) {
            cur += chunk_bytes[k];
            const uintptr_t hi = std::min(cur & ~(page - 1), last_page);
            regions.emplace_back(reinterpret_cast<char *>(lo), hi > lo ? (size_t)(hi - lo) : 0);
            lo = std::max(lo, hi);
        }
        planned = regions;
        worker = std::thread([this]() {
            (void)hipSetDevice(device);
            for (size_t k = 0; k < regions.size(); k++) {
                if (!failed.load() && regions[k].second &&
                    hipHostRegister(regions[k].first, regions[k].second, hipHostRegisterDefault) != hipSuccess) {
                    (void)hipGetLastError();
                    regions[k].second = 0;               // not ours to unregister
                    failed.store(true);
                } else if (failed.load()) {
                    regions[k].second = 0;
                }
                done.store((int)k + 1, std::memory_order_release);
            }
        });
    }
    // Wait until every region that overlaps [dst, dst + bytes) is locked (or locking has been given up), then return the
    // offsets at which a copy of that range has to be cut: every region boundary strictly inside it.
    std::vector<size_t> cuts_for(const char *dst, size_t bytes) const {
        std::vector<size_t> cuts;
        int need = 0;
        for (size_t k = 0; k < planned.size(); k++) {          // `planned` is immutable once the worker runs
            const char *lo = planned[k].first, *hi = lo + planned[k].second;
            if (planned[k].second && hi > dst && lo < dst + bytes) need = (int)k + 1;
        }
        while (done.load(std::memory_order_acquire) < need) std::this_thread::yield();
        for (int k = 0; k < need; k++) {                       // regions[k] is final for k < done (release / acquire)
            const char *lo = planned[(size_t)k].first, *hi = lo + planned[(size_t)k].second;
            if (planned[(size_t)k].second == 0) continue;
            if (lo > dst && lo < dst + bytes) cuts.push_back((size_t)(lo - dst));
            if (hi > dst && hi < dst + bytes) cuts.push_back((size_t)(hi - dst));
        }
        std::sort(cuts.begin(), cuts.end());
        cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
        return cuts;
    }
    void finish() {          // after every copy has completed
        if (worker.joinable()) worker.join();
        for (auto &r : regions) if (r.second) (void)hipHostUnregister(r.first);
        regions.clear();
    }
    ~HostPinner() { finish(); }
};

// The certificate monitor of opts.verify_near without count_work: see horizon_run.
struct NearMonitor {
    int device = 0;
    hipStream_t st_mon = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    void *buf = nullptr;
    size_t buf_bytes = 0;
    unsigned long long *cnt = nullptr;
    bool active = false;
    int launch(const Scene *sc, const HorizonArgs &a, int n, int seed, hipStream_t st) {
        device = sc->device;
        const int nb = horizon_num_blocks(a);
        std::vector<int> pick;
        for (int b = 0; b < nb; b++) {
            uint32_t h = (uint32_t)b * 2654435761u + (uint32_t)seed * 40503u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            if (h % (uint32_t)n == 0u) pick.push_back(b);
        }
        if (pick.empty()) return HZ_OK;
        int rc;
        if (!st_mon && (rc = stream_acquire(device, &st_mon))) return rc;
        if (!ready) { HZ_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming)); HZ_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming)); }
        const size_t cbytes = HZ_CNT_N * sizeof(unsigned long long) + HZ_REDO_CAP * sizeof(int);
        const size_t row_bytes = ((size_t)a.azim_num * sizeof(float) + 255) & ~(size_t)255;     // the scratch row its stores go to
        const size_t need = cbytes + row_bytes + pick.size() * sizeof(int);
        if (need > buf_bytes) {                               // (allocated once per call: the chunks of a call pick about as many blocks)
            if (buf) { (void)hipFree(buf); buf = nullptr; buf_bytes = 0; }
            HZ_HIP(hipMalloc(&buf, need + need / 4));
            buf_bytes = need + need / 4;
        }
        cnt = (unsigned long long *)buf;
        float *scratch = (float *)((char *)buf + cbytes);
        int *d_list = (int *)((char *)buf + cbytes + row_bytes);
        HZ_HIP(hipEventRecord(ready, st));                    // certificates (and everything enqueued before) are complete
        HZ_HIP(hipStreamWaitEvent(st_mon, ready, 0));
        HZ_HIP(hipMemsetAsync(buf, 0, cbytes, st_mon));
        HZ_HIP(hipMemcpyAsync(d_list, pick.data(), pick.size() * sizeof(int), hipMemcpyHostToDevice, st_mon));
        HZ_HIP(hipStreamSynchronize(st_mon));                 // `pick` is host memory of this scope; the copy is tiny
        HorizonArgs v = a;
        v.count_work = 1; v.verify_near = 1; v.level_stack = 1;
        v.scratch_row = scratch;                              // its stores must not land in the production launch's rows
        v.counters = cnt; v.tile_list = d_list; v.n_list = (int)pick.size();
        if ((rc = horizon_launch(sc, v, st_mon, nullptr))) return rc;
        HZ_HIP(hipEventRecord(done, st_mon));
        active = true;
        return HZ_OK;
    }
    // the production stream waits for the monitor (later kernels read the rows it rewrote); tallies are added up
    int collect(hipStream_t st, unsigned long long *verified, unsigned long long *violations) {
        unsigned long long c2[HZ_CNT_N] = {0};
        HZ_HIP(hipStreamWaitEvent(st, done, 0));
        HZ_HIP(hipMemcpyAsync(c2, cnt, sizeof(c2), hipMemcpyDeviceToHost, st_mon));
        HZ_HIP(hipStreamSynchronize(st_mon));
        *verified += c2[21]; *violations += c2[10];
        active = false;
        return HZ_OK;
    }
    ~NearMonitor() {
        if (st_mon) { (void)hipSetDevice(device); stream_release(device, st_mon); }
        if (ready) (void)hipEventDestroy(ready);
        if (done) (void)hipEventDestroy(done);
        if (buf) (void)hipFree(buf);
    }
};

// opts.verbose >= 2: why cells got no certificate (device histogram of hz_near.hip), to stderr
static void print_near_reasons(const unsigned *near_reasons) {
    unsigned h[20] = {0};
    if (hipMemcpy(h, near_reasons, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
    static const char *nm[14] = {"frame", "vertex_on_axis", "edge_over_axis", "az_tolerance", "inplane_edge", "crossing_near_axis",
                                 "interval", "precision", "edge_on", "orientation", "origin_below", "axis_in_triangle", "window_or_mask",
                                 "bad_mesh_nearby"};
    fprintf(stderr, "hz near reasons: cells %u certified %u", h[0], h[1]);
    for (int b = 0; b < 14; b++) if (h[2 + b]) fprintf(stderr, " %s %u", nm[b], h[2 + b]);
    fprintf(stderr, " | tasks %u bins %u\n", h[16], h[17]);
}

// opts.verbose: the reference's report, horizon_comp.cpp:673-700, 805-810 (same lines, same order)
static void print_reference_report(const Scene *sc, int alg, unsigned long long cells, unsigned long long rays, size_t slab_cells,
                                   int dim_in_0, int dim_in_1, int azim_num, double kernel_s, bool certificates_off, bool certificates_per_cell) {
    static const char *alg_name[3] = {"discrete_sampling", "binary search", "guess horizon from previous azimuth direction"};
    printf("Horizon detection algorithm: %s\n", alg_name[alg]);
    printf("Number of grid cells for which horizon is computed: %llu \n", cells);
    printf("Fraction of total number of grid cells: %g %%\n", (double)((float)cells / (float)slab_cells * 100.0f));
    printf("Total memory required for horizon output: %g GB\n",
           (double)(((float)dim_in_0 * (float)dim_in_1 * (float)azim_num * 4.0f) / 1.0e9f));
    printf("Ray tracing time: %g s\n", kernel_s);
    printf("Number of rays shot: %llu\n", rays);
    printf("Average number of rays per location and azimuth: %.2f \n",
           cells ? (double)((float)rays / (float)(cells * (unsigned long long)azim_num)) : 0.0);
    // (not a line of the reference: why this call ran without the near-field certificates, if it did)
    if (certificates_off)
        printf("Near-field certificates off: a DEM quad that is not part of a height field over the (x, y) plane, or a triangle "
               "of the outer simplified domain, is too large for the per-cell guard (results unaffected, slower)\n");
    else if (certificates_per_cell)
        printf("Near-field certificates per cell: %u DEM triangle(s) project onto the (x, y) plane collapsed or against the "
               "majority, %d outer-domain triangle(s); cells near them run without (results unaffected)\n", sc->hdr.n_flipped, sc->hdr.n_tin);
    fflush(stdout);
}

// Rows per launch and the device buffer of one chunk of horizon, when `hori` is host memory or skipped (SVF only).
// Every launch ends with a tail (the last waves run on a draining GPU; a lane owns its cell for all azimuths), so launches
// should be few: at least 16 GiB of horizon per launch when it is only the SVF's input (nothing is copied out; less if HBM is
// short), at least 4 GiB when chunks are copied to the host behind the next chunk's kernel.
static int alloc_hori_chunk(int rows_all, size_t row_bytes, bool skip_hori, int fixed_rows, int *chunk_rows_out, void **buf) {
    size_t target = skip_hori ? ((size_t)16 << 30) : ((size_t)4 << 30);
    {   // host path with plenty of free HBM: equal chunks of <= 8 GiB (the 3601^2 tile: 3 instead of 5 launches, 2.57 ->
        // 2.49 s).  (64 GiB chunks for the SVF-only path were measured on config 5: 5 instead of 18 launches save
        // 0.7 s of kernel tails and cost 2 s of hipMalloc / hipFree of the 62 + 33 GB buffers -- kept at 16 GiB.)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            if (!skip_hori) {
                const size_t all = (size_t)rows_all * row_bytes;
                const size_t parts = std::max<size_t>(1, (all + ((size_t)8 << 30) - 1) / ((size_t)8 << 30));
                if (free_b / 6 >= all / parts + row_bytes) target = std::max(target, all / parts + row_bytes);
            }
        } else (void)hipGetLastError();
    }
    const bool fixed = fixed_rows > 0;
    for (;;) {
        int chunk_rows = rows_all;
        if (fixed) chunk_rows = std::min(chunk_rows, fixed_rows);
        else {
            chunk_rows = (int)std::max<size_t>(16, std::min<size_t>((size_t)chunk_rows, target / row_bytes));
            // equal chunks (whole 16-row tile rows) instead of full ones and a small rest: a slab of 1800 rows is
            // 3 x 608 rows, not 830 + 830 + 140 -- every launch ends with a tail, and a short launch is mostly tail
            const int n_ch = (rows_all + chunk_rows - 1) / chunk_rows;
            chunk_rows = std::min(chunk_rows, std::max(16, ((rows_all + n_ch - 1) / n_ch + 15) / 16 * 16));
        }
        chunk_rows = std::min(chunk_rows, rows_all);
        *chunk_rows_out = chunk_rows;
        if (hipMalloc(buf, (size_t)chunk_rows * row_bytes) == hipSuccess) return HZ_OK;
        (void)hipGetLastError();
        *buf = nullptr;
        if (fixed || target <= ((size_t)1 << 30)) return set_error(HZ_ERR_HIP, "hipMalloc of the horizon chunk (%zu bytes) failed", (size_t)chunk_rows * row_bytes);
        target >>= 1;
    }
}

// Near-field certificates of the rows [na.row_begin, na.row_end) (hz_near.hip) into the scratch kept with the scene (grown on
// demand: (2 A + 4) B per cell); fills na.near_idx / near_r, adds the pre-pass's GPU time to *ms_near.
static int near_prepass(const Scene *sc, NearArgs &na, hipStream_t st, float *ms_near) {
    const size_t cells = (size_t)(na.row_end - na.row_begin) * na.dim_in_1;
    const size_t idx_bytes = (cells * (size_t)na.azim_num * 2 + 255) & ~(size_t)255;
    const size_t need = idx_bytes + cells * 4;
    if (sc->near_bytes < need) {
        if (sc->near_buf) (void)hipFree(sc->near_buf);
        sc->near_buf = nullptr; sc->near_bytes = 0;
        if (hipMalloc(&sc->near_buf, need) != hipSuccess) return set_error(HZ_ERR_HIP, "hipMalloc of the near-field certificates failed");
        sc->near_bytes = need;
    }
    na.near_idx = (unsigned short *)sc->near_buf;
    na.near_r = (float *)((char *)sc->near_buf + idx_bytes);
    hipEvent_t n0 = nullptr, n1 = nullptr;
    if (hipEventCreate(&n0) != hipSuccess || hipEventCreate(&n1) != hipSuccess) {
        if (n0) (void)hipEventDestroy(n0);
        return set_error(HZ_ERR_HIP, "hipEventCreate failed");
    }
    (void)hipEventRecord(n0, st);
    const int rc = near_launch(sc, na, st);
    (void)hipEventRecord(n1, st);
    if (!rc) {
        (void)hipEventSynchronize(n1);
        float mn = 0.0f;
        (void)hipEventElapsedTime(&mn, n0, n1);
        *ms_near += mn;
    }
    (void)hipEventDestroy(n0); (void)hipEventDestroy(n1);
    return rc;
}

// Leftover cells (hz_horizon.hip): a block of a production launch ends when at most t[0] (default 36; opts.left_min) of its cells are
// unfinished -- a lane whose cell is done idles until its block's slowest cell is (12 % of the lane time, profiles/r05/
// probe_done_lanes.log) -- and follow-up launches finish the cells handed over, 64 per wave, sorted by the azimuths they have left
// and by position (left_sort); a wave of follow-up launch l may hand over again at t[l], the last one runs to the end.  How many
// records a level got is only known on the device: keys, sort and follow-up launches are sized by the region's capacity and read
// the counts there, the host never waits between them.  One 64 B record per hand-over; region l is sized from what the level
// above can hand over at most; a region that still runs out of room stops the hand-over (hz_horizon.hip).  Kept with the scene.
// Fills left_t[] (thresholds per level) and a.left_base / left_cap / left_sort / left_cap_max, grows the scene's buffer; returns the
// number of follow-up launches (0: no hand-over -- switched off, a counting call, or no memory: the blocks run to their end).
static int plan_leftover(const Scene *sc, const hz_opts *opts, int rows_max, int dim_in_1, int left_t[HZ_LEFT_LEVELS], HorizonArgs &a) {
    int n_left_launches = 0;
    for (int l = 0; l < HZ_LEFT_LEVELS; l++) left_t[l] = 0;
    const int pack = opts ? opts->left_min : 0;
    const unsigned upack = pack == 0 ? HZ_LEFT_DEFAULT : (pack < 0 ? 0u : (unsigned)pack);
    for (int l = 0; l < HZ_LEFT_LEVELS; l++) {
        left_t[l] = (int)std::min((upack >> (8 * l)) & 0xffu, 56u);
        if (left_t[l] == 0) break;      // (a level that does not hand over is the last one)
        n_left_launches = l + 1;
    }
    if ((opts && opts->count_work) || n_left_launches == 0) return 0;
    const TileMap tm = make_tile_map((rows_max + 15) / 16, (dim_in_1 + 15) / 16);
    unsigned long long units = (unsigned long long)tm.per_xcd * 8ull * 4ull;      // 8 x 8 blocks, then groups of 64 records
    unsigned long long total = 0, cap_max = 0;
    for (int l = 0; l < n_left_launches; l++) {
        unsigned long long cap = ((units * (unsigned long long)left_t[l] + 63ull) & ~63ull) + 64ull;
        if (opts && opts->left_cap_test > 0) cap = std::min<unsigned long long>(cap, ((unsigned long long)opts->left_cap_test + 63ull) & ~63ull);
        a.left_base[l] = (unsigned)total; a.left_cap[l] = (unsigned)cap;
        total += cap; cap_max = std::max(cap_max, cap);
        units = cap / 64ull;
    }
    const size_t rec_bytes = (size_t)total * HZ_LEFT_WORDS * sizeof(unsigned);
    const size_t sort_words = 4 * (size_t)cap_max + sort_temp_elems((size_t)cap_max) + 64;
    const size_t need = rec_bytes + sort_words * sizeof(uint32_t);
    // (record numbers and launch-local cell numbers are 32 bit)
    if (total >= (1ull << 31) || (unsigned long long)rows_max * (unsigned long long)dim_in_1 >= 0xffffffffull) return 0;
    if (sc->left_bytes < need) {
        if (sc->left_buf) (void)hipFree(sc->left_buf);
        sc->left_buf = nullptr; sc->left_bytes = 0;
        if (hipMalloc(&sc->left_buf, need) != hipSuccess) { (void)hipGetLastError(); sc->left_buf = nullptr; return 0; }
        sc->left_bytes = need;
    }
    a.left_sort = (uint32_t *)((char *)sc->left_buf + rec_bytes); a.left_cap_max = (unsigned)cap_max;
    return n_left_launches;
}

static int horizon_run(const Scene *sc, const float *vec_norm, const float *vec_north, int offset_0,
                       int offset_1, float *hori_buffer, int dim_in_0, int dim_in_1, int azim_num,
                       float dist_search, float hori_acc, const char *ray_algorithm,
                       float elev_ang_low_lim, const uint8_t *mask, float hori_fill, float ray_org_elev,
                       const hz_opts *opts, hz_stats *stats) {
    int alg = 2;
    int rc = parse_alg(ray_algorithm, &alg);
    if (rc) return rc;
    if (!vec_norm || !vec_north || !mask) return set_error(HZ_ERR_ARG, "vec_norm, vec_north and mask must not be NULL");
    if (dim_in_0 <= 0 || dim_in_1 <= 0 || azim_num <= 0) return set_error(HZ_ERR_ARG, "dim_in_0, dim_in_1 and azim_num must be positive");
    if (offset_0 < 0 || offset_1 < 0 || offset_0 + dim_in_0 > sc->hdr.d0 || offset_1 + dim_in_1 > sc->hdr.d1)
        return set_error(HZ_ERR_ARG, "inconsistency between input arguments dem_dim_0, dem_dim_1, offset_0, offset_1 and vec_norm");
    if (!(hori_acc > 0.0f) || hori_acc > 10.0f) return set_error(HZ_ERR_ARG, "limit of hori_acc (10 degree) is exceeded");
    const bool skip_hori = opts && opts->skip_hori;
    if (!hori_buffer && !skip_hori) return set_error(HZ_ERR_ARG, "hori_buffer is NULL");
    if (opts && opts->svf && !opts->vec_tilt) return set_error(HZ_ERR_ARG, "opts.svf needs opts.vec_tilt");
    // the SVF weights sectors by azim[1] - azim[0] (topo_param.pyx:433): undefined for a single azimuth
    if (opts && opts->svf && azim_num < 2) return set_error(HZ_ERR_ARG, "opts.svf needs azim_num >= 2");
    // row slab (include/horayzon_hip.h): {0, 0} = the whole inner domain (a zeroed struct), row_end == -1 = dim_in_0,
    // anything else must satisfy 0 <= row_begin <= row_end <= dim_in_0; begin == end is an empty slab: nothing to do
    int row_begin = 0, row_end = dim_in_0;
    if (opts && !(opts->row_begin == 0 && opts->row_end == 0)) {
        row_begin = opts->row_begin;
        row_end = (opts->row_end == -1) ? dim_in_0 : opts->row_end;
        if (row_begin < 0 || row_end < 0 || row_end > dim_in_0 || row_begin > row_end)
            return set_error(HZ_ERR_ARG, "invalid row slab [%d, %d) for dim_in_0 = %d", opts->row_begin, opts->row_end, dim_in_0);
        if (row_begin == row_end) return HZ_OK;
    }
    HZ_HIP(hipSetDevice(sc->device));
    std::lock_guard<std::mutex> run_lock(sc->run_mu);     // one call at a time on a scene's stream and scratch
    hipStream_t st = sc->stream;
    Timer t_total; t_total.start();

    // unit conversions: horizon_comp.cpp:667-670
    HostTables tb;
    build_tables(azim_num, hori_acc, elev_ang_low_lim, tb);
    if (tb.elev_num < 2) return set_error(HZ_ERR_ARG, "elevation table is empty (elev_ang_low_lim too high)");
    const float dist_m = (float)((double)dist_search * 1000.0);

    const size_t ncell = (size_t)dim_in_0 * dim_in_1;
    const size_t slab_cells = (size_t)(row_end - row_begin) * dim_in_1;
    Timer t_h2d; t_h2d.start();
    DevIn<float> d_norm, d_north, d_tilt, d_as, d_ac, d_ea, d_es, d_ec;
    DevIn<int> d_mid;
    DevIn<uint8_t> d_mask;
    // Per-cell inputs: only the slab's rows are ever read, so only those are uploaded when the arrays are host memory.
    // opts.inputs_are_slab: the four pointers address row_begin (a rank of a sharded job holds -- and uploads -- its own
    // rows only); otherwise they address inner-domain row 0 (the reference's layout).  The kernels index by global
    // cell: `in_off` shifts the slab-local device addresses back (address arithmetic only, never dereferenced outside
    // the slab).
    const bool inputs_are_slab = opts && opts->inputs_are_slab;
    const size_t in_off = (size_t)row_begin * dim_in_1;
    const size_t src_off = inputs_are_slab ? 0 : in_off;
    (void)ncell;
    if ((rc = d_norm.bind(vec_norm + 3 * src_off, slab_cells * 3, st))) return rc;
    if ((rc = d_north.bind(vec_north + 3 * src_off, slab_cells * 3, st))) return rc;
    if ((rc = d_mask.bind(mask + src_off, slab_cells, st))) return rc;
    if (opts && opts->svf) if ((rc = d_tilt.bind(opts->vec_tilt + 3 * src_off, slab_cells * 3, st))) return rc;
    const float *norm0 = d_norm.dev - 3 * in_off, *north0 = d_north.dev - 3 * in_off;
    const uint8_t *mask0 = d_mask.dev - in_off;
    const float *tilt0 = d_tilt.dev ? d_tilt.dev - 3 * in_off : nullptr;
    if ((rc = d_as.bind(tb.azim_sin.data(), (size_t)azim_num, st))) return rc;
    if ((rc = d_ac.bind(tb.azim_cos.data(), (size_t)azim_num, st))) return rc;
    if ((rc = d_ea.bind(tb.elev_ang.data(), (size_t)tb.elev_num, st))) return rc;
    if ((rc = d_es.bind(tb.elev_sin.data(), (size_t)tb.elev_num, st))) return rc;
    if ((rc = d_ec.bind(tb.elev_cos.data(), (size_t)tb.elev_num, st))) return rc;
    if ((rc = d_mid.bind(tb.mid_idx.data(), tb.mid_idx.size(), st))) return rc;
    DevOut<float> d_hori, d_svf;
    DevIn<float> d_azim;
    const bool want_svf = opts && opts->svf;
    // hori_buffer addresses inner-domain row 0 (the reference's layout) unless opts.hori_is_slab says it addresses
    // row_begin.  Only the slab itself is ever classified or written: `hori_row0` is an address used for
    // arithmetic alone (it may lie below the caller's allocation when a resident slab buffer is passed).
    const bool hori_is_slab = opts && opts->hori_is_slab;
    float *hori_slab_host = hori_buffer ? (hori_is_slab ? hori_buffer : hori_buffer + (size_t)row_begin * dim_in_1 * azim_num) : nullptr;
    float *hori_row0 = hori_slab_host ? hori_slab_host - (size_t)row_begin * dim_in_1 * azim_num : nullptr;
    float *svf_slab = want_svf ? (hori_is_slab ? opts->svf : opts->svf + (size_t)row_begin * dim_in_1) : nullptr;
    if ((rc = d_svf.bind(svf_slab, svf_slab ? slab_cells : 0))) return rc;
    // rows per launch: the whole slab when `hori` is device memory.  Otherwise the horizon of a chunk
    // of rows lives in a bounded temporary: one <= 4 GiB buffer when only the SVF is wanted, two of
    // them when `hori` is host memory -- chunk c is copied out on a second stream while chunk c + 1 is
    // traced, so the D2H time hides behind the kernel and the output may be larger than HBM.
    int chunk_rows = row_end - row_begin;
    void *tmp_hori = nullptr, *tmp_hori2 = nullptr;
    size_t tmp_bytes = 0;         // chunk buffers of a host / skipped hori_buffer (hz_stats.scratch_bytes)
    struct TmpFree { void **p; ~TmpFree() { if (*p) (void)hipFree(*p); } } tmp_free{&tmp_hori}, tmp_free2{&tmp_hori2};
    const size_t row_bytes = (size_t)dim_in_1 * azim_num * 4;
    const bool stream_out = !skip_hori && !is_device_ptr(hori_slab_host);
    if (skip_hori || stream_out) {
        if (skip_hori && !want_svf) return set_error(HZ_ERR_ARG, "skip_hori without svf: nothing to compute");
        if ((rc = alloc_hori_chunk(row_end - row_begin, row_bytes, skip_hori, (opts && opts->chunk_rows > 0) ? opts->chunk_rows : 0, &chunk_rows, &tmp_hori)))
            return rc;
        tmp_bytes = (size_t)chunk_rows * row_bytes;
        if (stream_out && chunk_rows < row_end - row_begin) { HZ_HIP(hipMalloc(&tmp_hori2, (size_t)chunk_rows * row_bytes)); tmp_bytes *= 2; }
    } else {
        if ((rc = d_hori.bind(hori_slab_host, slab_cells * (size_t)azim_num))) return rc;
    }
    std::vector<float> azim_h;
    if (want_svf) {   // azim as the wrapper recomputes it, horizon.pyx:191-195
        azim_h.resize((size_t)azim_num);
        for (int i = 0; i < azim_num; i++) azim_h[(size_t)i] = (float)(((2 * M_PI) / azim_num) * i);
        if ((rc = d_azim.bind(azim_h.data(), (size_t)azim_num, st))) return rc;
    }
    DevIn<unsigned long long> d_cnt;
    unsigned long long zeros[HZ_CNT_N] = {0};      // counters; behind them the list of tiles to redo (hz_horizon.hip)
    void *cnt_dev = nullptr;
    HZ_HIP(hipMalloc(&cnt_dev, sizeof(zeros) + 2 * HZ_REDO_CAP * sizeof(int)));      // (two lists: blocks, and groups of the leftover launch)
    d_cnt.owned = cnt_dev;
    HZ_HIP(hipStreamSynchronize(st));
    const double h2d_s = t_h2d.stop();

    HorizonArgs a;
    a.vec_norm = norm0; a.vec_north = north0; a.mask = mask0;
    a.offset_0 = offset_0; a.offset_1 = offset_1; a.dim_in_0 = dim_in_0; a.dim_in_1 = dim_in_1;
    a.azim_num = azim_num; a.elev_num = tb.elev_num; a.alg = alg;
    a.hori_acc = tb.hori_acc; a.low = tb.low; a.up = tb.up; a.dist = dist_m;
    a.hori_fill = hori_fill; a.ray_org_elev = ray_org_elev;
    a.azim_sin = d_as.dev; a.azim_cos = d_ac.dev; a.elev_ang = d_ea.dev; a.elev_sin = d_es.dev; a.elev_cos = d_ec.dev;
    a.mid_idx = d_mid.dev;
    a.top_nodes = opts ? opts->top_nodes : -1;
    a.regroup = opts ? opts->regroup : -1;
    a.count_work = opts ? opts->count_work : 0;
    a.hit_cache = (opts && opts->no_hit_cache) ? 0 : 1;
    a.counters = (unsigned long long *)cnt_dev;
    // near-field certificates (hz_near.hip): one pre-pass per chunk into a scratch buffer kept with the scene.
    // Off beyond the azimuth count the pre-pass holds in LDS, and on request.  The distance bound near_r assumes a height
    // field over the world (x, y) plane (HZ_BLOB_HEIGHT_FIELD, checked by the scene build).  Where only SOME quads break
    // that, or an outer-domain TIN is present, the scene carries a bitmap of their (x, y) footprints (HZ_BLOB_BAD_MAP) and
    // the pre-pass refuses the cells near them, one by one (round 5; rounds 3-4: off for the whole scene).  Neither flag
    // (a bad primitive too large to rasterise): off.  opts.no_near_skip < 0 overrides both checks (tests only).
    const int near_opt = opts ? opts->no_near_skip : 0;
    const bool height_field = (sc->hdr.flags & HZ_BLOB_HEIGHT_FIELD) != 0;
    const bool bad_map = (sc->hdr.flags & HZ_BLOB_BAD_MAP) != 0;
    const bool use_near = near_opt <= 0 && (height_field || bad_map || near_opt < 0) &&
                          azim_num <= near_max_azim() && tb.elev_num <= 65534;
    a.near_idx = nullptr; a.near_r = nullptr; a.scratch_row = nullptr;
    a.tile_list = nullptr; a.n_list = 0;
    // opts.verify_near = N: with count_work the counting instantiation re-traces one of every N shortened rays; without it
    // the production launch stays as it is and a second, counting launch re-traces EVERY shortened ray of one of every N
    // 8 x 8 blocks (the monitor for production inputs)
    const int verify_n = (opts && opts->verify_near > 0) ? opts->verify_near : 0;
    a.verify_near = (opts && opts->count_work) ? verify_n : 0;
    float ms_near = 0.0f;

    unsigned long long cnt[16] = {0}, n_verified = 0, n_mon_violations = 0;
    NearMonitor mon;
    float ms = 0.0f, ms_svf = 0.0f;
    int fallbacks = 0;
    unsigned long long redo_blocks = 0, redo_groups = 0;

    // HIP events on the kernels' stream: horizon kernel and SVF kernel are timed separately
    struct Ev { hipEvent_t a = nullptr, b = nullptr, c = nullptr, d = nullptr, l = nullptr; };
    std::vector<Ev> evs;          // one per launch attempt
    std::vector<size_t> ev_of;    // chunk -> its final attempt
    hipStream_t st_copy = nullptr;
    auto free_events = [&]() {
        for (auto &e : evs) {
            if (e.a) (void)hipEventDestroy(e.a);
            if (e.b) (void)hipEventDestroy(e.b);
            if (e.c) (void)hipEventDestroy(e.c);
            if (e.d) (void)hipEventDestroy(e.d);
            if (e.l) (void)hipEventDestroy(e.l);
        }
        if (st_copy) { stream_release(sc->device, st_copy); st_copy = nullptr; }
    };
    if (stream_out && (rc = stream_acquire(sc->device, &st_copy))) return rc;
    if (use_near) {             // the certificate scratch of the largest chunk, allocated before the pinner thread starts
        const size_t cells = (size_t)std::min(chunk_rows, row_end - row_begin) * dim_in_1;      // (page locking and hipMalloc
        const size_t need = ((cells * (size_t)azim_num * 2 + 255) & ~(size_t)255) + cells * 4;  //  serialise in the driver)
        if (sc->near_bytes < need) {
            if (sc->near_buf) (void)hipFree(sc->near_buf);
            sc->near_buf = nullptr; sc->near_bytes = 0;
            if (hipMalloc(&sc->near_buf, need) != hipSuccess) { free_events(); return set_error(HZ_ERR_HIP, "hipMalloc of the near-field certificates failed"); }
            sc->near_bytes = need;
        }
    }
    int left_t[HZ_LEFT_LEVELS];
    const int n_left_launches = plan_leftover(sc, opts, std::min(chunk_rows, row_end - row_begin), dim_in_1, left_t, a);
    a.left_min = n_left_launches > 0 ? left_t[0] : 0;
    a.left_rec = n_left_launches > 0 ? (unsigned *)sc->left_buf : nullptr;
    a.left_regroup = (opts && (opts->left_tune & 0xff) > 0) ? (opts->left_tune & 0xff) : HZ_LEFT_REGROUP;
    a.left_key_shift = (opts && ((opts->left_tune >> 8) & 0xff) > 0) ? std::min(((opts->left_tune >> 8) & 0xff) - 1, 8) : HZ_LEFT_KEY_SHIFT;
    a.persist_grid = (opts && opts->persist_grid > 0) ? opts->persist_grid : 0;
    a.no_persist = (opts && opts->persist_grid < 0) ? 1 : 0;
    // opts.verbose >= 2: histogram of why cells got no certificate, printed to stderr at the end of the call
    unsigned *near_reasons = nullptr;
    struct ReasonsFree { unsigned **p; ~ReasonsFree() { if (*p) (void)hipFree(*p); } } reasons_free{&near_reasons};
    if (use_near && opts && opts->verbose >= 2) {
        if (hipMalloc((void **)&near_reasons, 20 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); near_reasons = nullptr; }
        else (void)hipMemsetAsync(near_reasons, 0, 20 * sizeof(unsigned), st);
    }
    HostPinner pinner;          // declared after the events: destroyed (joined, unregistered) before them
    if (stream_out && !(opts && opts->no_host_pin)) {
        std::vector<size_t> cb;
        for (int rb = row_begin; rb < row_end; rb += chunk_rows) cb.push_back((size_t)(std::min(rb + chunk_rows, row_end) - rb) * row_bytes);
        pinner.start(sc->device, reinterpret_cast<char *>(hori_slab_host), cb);
    }
    // copy of chunk k (issued after chunk k + 1 was launched, so a host-blocking pageable copy still overlaps)
    auto copy_out = [&](int k) -> int {
        const int rb = row_begin + k * chunk_rows, re = std::min(rb + chunk_rows, row_end);
        const void *src = (k & 1) ? tmp_hori2 : tmp_hori;
        Ev &e = evs[ev_of[(size_t)k]];
        char *dst = reinterpret_cast<char *>(hori_row0 + (size_t)rb * dim_in_1 * azim_num);
        const size_t bytes = (size_t)(re - rb) * row_bytes;
        // a copy must not straddle two separately page-locked ranges (nor a locked and a pageable one): cut it at every
        // region boundary inside the chunk (at most three pieces; the slab's unaligned head and tail stay pageable)
        std::vector<size_t> cuts;
        if (!pinner.planned.empty()) cuts = pinner.cuts_for(dst, bytes);
        cuts.push_back(bytes);
        if (hipStreamWaitEvent(st_copy, e.c, 0) != hipSuccess)
            return set_error(HZ_ERR_HIP, "copy of the horizon chunk failed: %s", hipGetErrorString(hipGetLastError()));
        size_t from = 0;
        for (size_t to : cuts) {
            if (to > from && hipMemcpyAsync(dst + from, static_cast<const char *>(src) + from, to - from, hipMemcpyDeviceToHost, st_copy) != hipSuccess)
                return set_error(HZ_ERR_HIP, "copy of the horizon chunk failed: %s", hipGetErrorString(hipGetLastError()));
            from = to;
        }
        if (hipEventRecord(e.d, st_copy) != hipSuccess)
            return set_error(HZ_ERR_HIP, "copy of the horizon chunk failed: %s", hipGetErrorString(hipGetLastError()));
        return HZ_OK;
    };
    auto fail = [&](int code) { (void)hipStreamSynchronize(st); if (st_copy) (void)hipStreamSynchronize(st_copy); free_events(); return code; };
    int n_chunk = 0;
    unsigned long long left_cells = 0, left_again = 0;
    float ms_left = 0.0f;
    for (int rb = row_begin; rb < row_end; rb += chunk_rows, n_chunk++) {
        const int re = std::min(rb + chunk_rows, row_end);
        // the kernels index hori by global cell: shift the (slab- or chunk-local) buffer back
        float *hori_chunk;
        if (stream_out) hori_chunk = (float *)((n_chunk & 1) ? tmp_hori2 : tmp_hori);
        else if (skip_hori) hori_chunk = (float *)tmp_hori;
        else hori_chunk = d_hori.dev + (size_t)(rb - row_begin) * dim_in_1 * azim_num;
        a.hori = hori_chunk - (size_t)rb * dim_in_1 * azim_num;
        a.row_begin = rb; a.row_end = re;
        if (use_near) {
            NearArgs na;
            na.vec_norm = norm0; na.vec_north = north0; na.mask = mask0;
            na.azim_sin = d_as.dev; na.azim_cos = d_ac.dev;
            na.offset_0 = offset_0; na.offset_1 = offset_1; na.dim_in_1 = dim_in_1; na.row_begin = rb; na.row_end = re;
            na.azim_num = azim_num; na.elev_num = tb.elev_num;
            na.ray_org_elev = ray_org_elev; na.hori_acc = tb.hori_acc; na.low = tb.low; na.up = tb.up;
            na.reasons = near_reasons;
            na.ignore_bad_map = near_opt < 0 ? 1 : 0;
            if ((rc = near_prepass(sc, na, st, &ms_near))) return fail(rc);
            a.near_idx = na.near_idx; a.near_r = na.near_r;
        }
        {
            Ev e;
            if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess || hipEventCreate(&e.c) != hipSuccess ||
                hipEventCreate(&e.d) != hipSuccess) {
                evs.push_back(e);
                return fail(set_error(HZ_ERR_HIP, "hipEventCreate failed"));
            }
            evs.push_back(e);
            ev_of.push_back(evs.size() - 1);
            if (hipMemcpyAsync(cnt_dev, zeros, sizeof(zeros), hipMemcpyHostToDevice, st) != hipSuccess)
                return fail(set_error(HZ_ERR_HIP, "counter reset failed"));
            // the buffer of chunk n is the one chunk n - 2 was copied out of
            if (stream_out && n_chunk >= 2) (void)hipStreamWaitEvent(st, evs[ev_of[(size_t)n_chunk - 2]].d, 0);
            a.level_stack = (sc->level_stack.load(std::memory_order_relaxed) != 0 || (opts && opts->level_stack > 0)) ? 1
                            : ((opts && opts->level_stack < 0) ? opts->level_stack : 0);
            (void)hipEventRecord(e.a, st);
            if (verify_n > 0 && !a.count_work && use_near) {
                // monitor: the certificates of a sample of this launch's blocks, checked ray by ray by a counting launch
                // (every shortened ray traced a second time over its full length) on a stream of its own, so that its few
                // workgroups run NEXT TO the production launch instead of after it (a launch of its own costs one
                // workgroup lifetime, ~0.1 s, however small the sample).  Its stores go to a scratch row (HorizonArgs::scratch_row).
                if ((rc = mon.launch(sc, a, verify_n, rb, st))) return fail(rc);
            }
            // production launch + the follow-up launches of the cells it leaves unfinished (no host wait in between)
            auto launch_chunk = [&](int *safe_out) -> int {
                int r = horizon_launch(sc, a, st, safe_out);
                if (r || n_left_launches == 0) return r;
                if (!evs.back().l && hipEventCreate(&evs.back().l) != hipSuccess) return set_error(HZ_ERR_HIP, "hipEventCreate failed");
                (void)hipEventRecord(evs.back().l, st);
                for (int l = 1; l <= n_left_launches && !r; l++) {
                    if ((r = left_sort(a, l - 1, st))) break;
                    HorizonArgs b = a;
                    b.left_mode = l; b.left_min = l < n_left_launches ? left_t[l] : 0;
                    // a single follow-up launch runs the stack discipline of the production launch (a group whose fast stack overflows is
                    // computed again, below, from the sorted order this launch used); with several levels they run one entry per level:
                    // nothing can overflow there (and the next level's sort reuses the buffers of this one's)
                    if (n_left_launches > 1) b.level_stack = 1;
                    b.tile_list = nullptr; b.n_list = 0;
                    r = horizon_launch(sc, b, st, nullptr);
                }
                return r;
            };
            int safe = 0;
            rc = launch_chunk(&safe);
            if (!rc && !safe) {
                // fast stack discipline: did a wave run out of entries?  Its 8 x 8 block does not count (and handed nothing over) and is
                // computed again with the one-entry-per-level kernel: block by block when they are few (deep trees overflow in a
                // few places only), the whole launch -- and every later launch on this scene -- when they are many
                unsigned long long cov[32] = {0};
                if (hipMemcpyAsync(cov, cnt_dev, sizeof(cov), hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipStreamSynchronize(st) != hipSuccess)
                    return fail(set_error(HZ_ERR_HIP, "horizon kernel failed: %s", hipGetErrorString(hipGetLastError())));
                const unsigned long long ov = cov[8], lov = cov[30];      // blocks / groups of the leftover launch that ran out of entries
                if (ov != 0 || lov != 0) {
                    const unsigned long long tiles = (unsigned long long)((re - rb + 7) / 8) * (unsigned long long)((dim_in_1 + 7) / 8);   // 8 x 8 blocks
                    fallbacks++;
                    a.level_stack = 1;
                    if (ov <= HZ_REDO_CAP && lov <= HZ_REDO_CAP && (ov + lov) * 4 <= tiles) {
                        redo_blocks += ov + lov; redo_groups += lov;
                        if (ov != 0) {
                            a.tile_list = reinterpret_cast<const int *>((unsigned long long *)cnt_dev + HZ_CNT_N);
                            a.n_list = (int)ov;
                            rc = horizon_launch(sc, a, st, &safe);
                            a.tile_list = nullptr; a.n_list = 0;
                        }
                        if (!rc && lov != 0) {
                            HorizonArgs b = a;
                            b.left_mode = 1; b.left_min = 0;
                            b.tile_list = reinterpret_cast<const int *>((unsigned long long *)cnt_dev + HZ_CNT_N) + HZ_REDO_CAP;
                            b.n_list = (int)lov;
                            rc = horizon_launch(sc, b, st, nullptr);
                        }
                    } else {
                        sc->level_stack.store(1, std::memory_order_relaxed);
                        if (hipMemcpyAsync(cnt_dev, zeros, sizeof(zeros), hipMemcpyHostToDevice, st) != hipSuccess)
                            return fail(set_error(HZ_ERR_HIP, "counter reset failed"));
                        (void)hipEventRecord(e.a, st);
                        rc = launch_chunk(&safe);
                    }
                }
            }
            if (!rc && mon.active) rc = mon.collect(st, &n_verified, &n_mon_violations);
            (void)hipEventRecord(e.b, st);
            if (!rc && want_svf)
                rc = svf_launch(d_azim.dev, hori_chunk, tilt0 + 3 * (size_t)rb * dim_in_1, re - rb, dim_in_1,
                                azim_num, d_svf.dev + (size_t)(rb - row_begin) * dim_in_1, st);
            (void)hipEventRecord(e.c, st);
            if (!rc && stream_out && n_chunk >= 1) rc = copy_out(n_chunk - 1);
            if (rc) return fail(rc);
            unsigned long long c[HZ_CNT_N];
            if (hipMemcpyAsync(c, cnt_dev, sizeof(c), hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)
                return fail(set_error(HZ_ERR_HIP, "horizon kernel failed: %s", hipGetErrorString(hipGetLastError())));
            float m1 = 0.0f, m2 = 0.0f;
            (void)hipEventElapsedTime(&m1, e.a, e.b);
            (void)hipEventElapsedTime(&m2, e.b, e.c);
            ms += m1; ms_svf += m2;
            if (evs.back().l) { float m3 = 0.0f; (void)hipEventElapsedTime(&m3, evs.back().l, e.b); ms_left += m3; }
            for (int k = 0; k < 16; k++) cnt[k] += c[k];
            left_cells += c[28]; left_again += c[29];
            n_verified += c[21];
            if (a.count_work && opts && opts->verbose >= 3) {      // per-XCD span of this launch (counting instantiation)
                const unsigned long long t0 = ~c[20];
                fprintf(stderr, "hz xcd spans [ms] rows %d..%d:", rb, re);
                for (int x = 0; x < 8; x++) fprintf(stderr, " %.1f", c[12 + x] ? (double)(c[12 + x] - t0) * 1.0e-5 : 0.0);
                fprintf(stderr, "\n");
            }
        }
    }
    Timer t_d2h; t_d2h.start();
    if (stream_out) {
        rc = copy_out(n_chunk - 1);
        const hipError_t se = hipStreamSynchronize(st_copy);
        pinner.finish();
        if (!rc && se != hipSuccess) rc = set_error(HZ_ERR_HIP, "copy of the horizon failed: %s", hipGetErrorString(se));
        if (rc) { free_events(); return rc; }
    }
    free_events();
    HZ_HIP(hipGetLastError());
    if ((rc = d_hori.finish(st))) return rc;
    if ((rc = d_svf.finish(st))) return rc;
    HZ_HIP(hipStreamSynchronize(st));
    const double d2h_s = t_d2h.stop();

    if (stats) {
        stats->num_rays += cnt[0]; stats->guard_events += cnt[1];
        stats->nodes_visited += cnt[2]; stats->tris_tested += cnt[3]; stats->num_cells += cnt[4];
        stats->wave_node_iters += cnt[5]; stats->wave_leaf_iters += cnt[6]; stats->wave_refills += cnt[7];
        stats->t_h2d_s += h2d_s; stats->t_kernel_s += (double)ms * 1e-3; stats->t_svf_s += (double)ms_svf * 1e-3;
        stats->t_d2h_s += d2h_s;
        stats->t_total_s += t_total.stop();
        stats->elev_num = tb.elev_num; stats->bvh_height = sc->hdr.height; stats->scene_bytes = sc->hdr.total_bytes;
        stats->stack_fallbacks += (uint64_t)fallbacks; stats->stack_redo_blocks += redo_blocks; stats->left_redo_groups += redo_groups;
        stats->rays_shortened += cnt[9]; stats->near_violations += cnt[10] + n_mon_violations; stats->t_near_s += (double)ms_near * 1e-3;
        stats->guard_cells += cnt[11]; stats->near_verified += n_verified;
        stats->t_left_s += (double)ms_left * 1e-3; stats->left_cells += left_cells; stats->left_again += left_again;
        stats->scratch_bytes = (uint64_t)((use_near ? sc->near_bytes : 0) + (n_left_launches > 0 ? sc->left_bytes : 0) + tmp_bytes);
        stats->height_field = height_field ? 1 : 0; stats->near_used = use_near ? 1 : 0;
    }
    if (near_reasons) print_near_reasons(near_reasons);
    if (opts && opts->verbose)
        print_reference_report(sc, alg, cnt[4], cnt[0], slab_cells, dim_in_0, dim_in_1, azim_num, (double)ms * 1e-3,
                               !use_near && near_opt <= 0 && !height_field && !bad_map, use_near && bad_map);
    return HZ_OK;
}

static int locations_run(const Scene *sc, const float *coords, const float *vec_norm, const float *vec_north,
                         float *hori_buffer, float *hori_dist_buffer, int num_loc, int azim_num,
                         float dist_search, float hori_acc, const char *ray_algorithm, float elev_ang_low_lim,
                         const float *ray_org_elev, int hori_dist_out, const hz_opts *opts, hz_stats *stats) {
    int alg = 1;
    int rc = parse_alg(ray_algorithm, &alg);
    if (rc) return rc;
    if (hori_dist_out && alg == 2)
        return set_error(HZ_ERR_ARG, "horizon detection algorithm 'guess_constant' not implemented for horizon "
                                     "distance computation");
    if (!coords || !vec_norm || !vec_north || !ray_org_elev || !hori_buffer) return set_error(HZ_ERR_ARG, "NULL argument");
    if (hori_dist_out && !hori_dist_buffer) return set_error(HZ_ERR_ARG, "hori_dist_buffer is NULL");
    if (num_loc <= 0 || azim_num <= 0) return set_error(HZ_ERR_ARG, "num_loc and azim_num must be positive");
    if (!(hori_acc > 0.0f) || hori_acc > 10.0f) return set_error(HZ_ERR_ARG, "limit of hori_acc (10 degree) is exceeded");
    HZ_HIP(hipSetDevice(sc->device));
    std::lock_guard<std::mutex> run_lock(sc->run_mu);
    hipStream_t st = sc->stream;
    Timer t_total; t_total.start();
    HostTables tb;
    build_tables(azim_num, hori_acc, elev_ang_low_lim, tb);
    if (tb.elev_num < 2) return set_error(HZ_ERR_ARG, "elevation table is empty (elev_ang_low_lim too high)");
    Timer t_h2d; t_h2d.start();
    DevIn<float> d_co, d_norm, d_north, d_roe, d_as, d_ac, d_ea, d_es, d_ec;
    DevIn<int> d_mid;
    const size_t n = (size_t)num_loc;
    if ((rc = d_co.bind(coords, n * 3, st))) return rc;
    if ((rc = d_norm.bind(vec_norm, n * 3, st))) return rc;
    if ((rc = d_north.bind(vec_north, n * 3, st))) return rc;
    if ((rc = d_roe.bind(ray_org_elev, n, st))) return rc;
    if ((rc = d_as.bind(tb.azim_sin.data(), (size_t)azim_num, st))) return rc;
    if ((rc = d_ac.bind(tb.azim_cos.data(), (size_t)azim_num, st))) return rc;
    if ((rc = d_ea.bind(tb.elev_ang.data(), (size_t)tb.elev_num, st))) return rc;
    if ((rc = d_es.bind(tb.elev_sin.data(), (size_t)tb.elev_num, st))) return rc;
    if ((rc = d_ec.bind(tb.elev_cos.data(), (size_t)tb.elev_num, st))) return rc;
    if ((rc = d_mid.bind(tb.mid_idx.data(), tb.mid_idx.size(), st))) return rc;
    // outputs are read-modify-write (untouched rows keep the caller's NaN): copy them in first
    DevOut<float> d_hori, d_dist;
    if ((rc = d_hori.bind(hori_buffer, n * (size_t)azim_num))) return rc;
    if (d_hori.host) HZ_HIP(hipMemcpyAsync(d_hori.dev, hori_buffer, n * (size_t)azim_num * 4, hipMemcpyHostToDevice, st));
    if (hori_dist_out) {
        if ((rc = d_dist.bind(hori_dist_buffer, n * (size_t)azim_num))) return rc;
        if (d_dist.host) HZ_HIP(hipMemcpyAsync(d_dist.dev, hori_dist_buffer, n * (size_t)azim_num * 4, hipMemcpyHostToDevice, st));
    }
    DevIn<unsigned long long> d_cnt;
    unsigned long long zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void *cnt_dev = nullptr;
    HZ_HIP(hipMalloc(&cnt_dev, sizeof(zeros)));
    d_cnt.owned = cnt_dev;
    HZ_HIP(hipMemcpyAsync(cnt_dev, zeros, sizeof(zeros), hipMemcpyHostToDevice, st));
    HZ_HIP(hipStreamSynchronize(st));
    const double h2d_s = t_h2d.stop();

    LocationsArgs a;
    a.coords = d_co.dev; a.vec_norm = d_norm.dev; a.vec_north = d_north.dev; a.ray_org_elev = d_roe.dev;
    a.hori = d_hori.dev; a.dist = d_dist.dev;
    a.num_loc = num_loc; a.azim_num = azim_num; a.elev_num = tb.elev_num; a.alg = alg; a.hori_dist_out = hori_dist_out;
    a.hori_acc = tb.hori_acc; a.low = tb.low; a.up = tb.up; a.dist_m = (float)((double)dist_search * 1000.0);
    a.azim_sin = d_as.dev; a.azim_cos = d_ac.dev; a.elev_ang = d_ea.dev; a.elev_sin = d_es.dev; a.elev_cos = d_ec.dev;
    a.mid_idx = d_mid.dev;
    a.counters = (unsigned long long *)cnt_dev;
    hipEvent_t e0, e1;
    HZ_HIP(hipEventCreate(&e0)); HZ_HIP(hipEventCreate(&e1));
    (void)hipEventRecord(e0, st);
    rc = locations_launch(sc, a, st);
    (void)hipEventRecord(e1, st);
    const hipError_t se = hipStreamSynchronize(st);
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (se != hipSuccess) return set_error(HZ_ERR_HIP, "locations kernel failed: %s", hipGetErrorString(se));
    Timer t_d2h; t_d2h.start();
    unsigned long long cnt[8];
    HZ_HIP(hipMemcpyAsync(cnt, cnt_dev, sizeof(cnt), hipMemcpyDeviceToHost, st));
    if ((rc = d_hori.finish(st))) return rc;
    if ((rc = d_dist.finish(st))) return rc;
    HZ_HIP(hipStreamSynchronize(st));
    if (stats) {
        stats->num_rays += cnt[0]; stats->guard_events += cnt[1]; stats->num_cells += cnt[4];
        stats->t_h2d_s += h2d_s; stats->t_kernel_s += (double)ms * 1e-3; stats->t_d2h_s += t_d2h.stop();
        stats->t_total_s += t_total.stop();
        stats->elev_num = tb.elev_num; stats->bvh_height = sc->hdr.height; stats->scene_bytes = sc->hdr.total_bytes;
    }
    if (opts && opts->verbose) {
        printf("Number of locations for which horizon is computed: %d \n", num_loc);
        printf("Ray tracing time: %g s\n", (double)ms * 1e-3);
        printf("Number of rays shot: %llu\n", cnt[0]);
    }
    return HZ_OK;
}

}  // namespace hz

using namespace hz;

extern "C" {

const char *hz_last_error(void) { return g_error.c_str(); }

int hz_abi_version(void) { return 6; }

int hz_abi_struct_sizes(int *opts_bytes, int *stats_bytes) {
    if (opts_bytes) *opts_bytes = (int)sizeof(hz_opts);
    if (stats_bytes) *stats_bytes = (int)sizeof(hz_stats);
    return HZ_OK;
}

int hz_device_count(int *count) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    if (count) *count = n;
    return HZ_OK;
}

int hz_device_info(int device, char *name, int cap, int *cu, uint64_t *hbm_bytes) {
    int rc = select_device(device);
    if (rc) return rc;
    hipDeviceProp_t prop;
    HZ_HIP(hipGetDeviceProperties(&prop, device));
    if (name && cap > 0) { strncpy(name, prop.gcnArchName, (size_t)cap - 1); name[cap - 1] = 0; }
    if (cu) *cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)prop.totalGlobalMem;
    return HZ_OK;
}

int hz_scene_create(const float *vert_grid, int dem_dim_0, int dem_dim_1, const char *geom_type,
                    const float *vert_simp, int num_vert_simp, const int32_t *tri_ind_simp,
                    int num_tri_simp, int device, hz_scene **scene, hz_stats *stats) {
    if (!scene || !vert_grid) return set_error(HZ_ERR_ARG, "scene / vert_grid is NULL");
    int rc = check_geom(geom_type);
    if (rc) return rc;
    Scene *sc = nullptr;
    if ((rc = scene_new(device, &sc))) return rc;
    rc = scene_build(sc, vert_grid, dem_dim_0, dem_dim_1, vert_simp, num_vert_simp, tri_ind_simp,
                     num_tri_simp, stats);
    if (rc) { scene_free(sc); return rc; }
    *scene = reinterpret_cast<hz_scene *>(sc);
    return HZ_OK;
}

int hz_scene_blob(const hz_scene *scene, void **device_ptr, size_t *nbytes) {
    if (!scene) return set_error(HZ_ERR_ARG, "scene is NULL");
    const Scene *sc = reinterpret_cast<const Scene *>(scene);
    if (device_ptr) *device_ptr = sc->blob;
    if (nbytes) *nbytes = sc->blob_bytes;
    return HZ_OK;
}

int hz_scene_vertices(const hz_scene *scene, const float **device_ptr, int *dem_dim_0, int *dem_dim_1, int *height_field) {
    if (!scene) return set_error(HZ_ERR_ARG, "scene is NULL");
    const Scene *sc = reinterpret_cast<const Scene *>(scene);
    if (device_ptr) *device_ptr = sc->verts();
    if (dem_dim_0) *dem_dim_0 = sc->hdr.d0;
    if (dem_dim_1) *dem_dim_1 = sc->hdr.d1;
    if (height_field) *height_field = (sc->hdr.flags & HZ_BLOB_HEIGHT_FIELD) ? 1 : 0;
    return HZ_OK;
}

int hz_scene_adopt(void *device_ptr, size_t nbytes, int device, hz_scene **scene) {
    if (!device_ptr || !scene || nbytes < sizeof(BlobHeader)) return set_error(HZ_ERR_ARG, "invalid blob");
    Scene *sc = nullptr;
    int rc = scene_new(device, &sc);
    if (rc) return rc;
    BlobHeader h;
    const hipError_t e = hipMemcpy(&h, device_ptr, sizeof(h), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { scene_free(sc); return set_error(HZ_ERR_HIP, "reading blob header failed: %s", hipGetErrorString(e)); }
    if (h.magic != HZ_BLOB_MAGIC || h.version != HZ_BLOB_VERSION || h.total_bytes > nbytes) {
        scene_free(sc);
        return set_error(HZ_ERR_ARG, "not a horayzon scene blob (magic/version/size mismatch)");
    }
    sc->blob = device_ptr; sc->blob_bytes = nbytes; sc->owns_blob = false; sc->hdr = h;
    *scene = reinterpret_cast<hz_scene *>(sc);
    return HZ_OK;
}

int hz_scene_destroy(hz_scene *scene) {
    scene_free(reinterpret_cast<Scene *>(scene));
    return HZ_OK;
}

int hz_horizon_gridded_scene(const hz_scene *scene, const float *vec_norm, const float *vec_north,
                             int offset_0, int offset_1, float *hori_buffer, int dim_in_0, int dim_in_1,
                             int azim_num, float dist_search, float hori_acc, const char *ray_algorithm,
                             float elev_ang_low_lim, const uint8_t *mask, float hori_fill,
                             float ray_org_elev, const hz_opts *opts, hz_stats *stats) {
    if (!scene) return set_error(HZ_ERR_ARG, "scene is NULL");
    return horizon_run(reinterpret_cast<const Scene *>(scene), vec_norm, vec_north, offset_0, offset_1,
                       hori_buffer, dim_in_0, dim_in_1, azim_num, dist_search, hori_acc, ray_algorithm,
                       elev_ang_low_lim, mask, hori_fill, ray_org_elev, opts, stats);
}

int hz_horizon_gridded(const float *vert_grid, int dem_dim_0, int dem_dim_1, const float *vec_norm,
                       const float *vec_north, int offset_0, int offset_1, float *hori_buffer,
                       int dim_in_0, int dim_in_1, int azim_num, float dist_search, float hori_acc,
                       const char *ray_algorithm, const char *geom_type, const float *vert_simp,
                       int num_vert_simp, const int32_t *tri_ind_simp, int num_tri_simp,
                       float elev_ang_low_lim, const uint8_t *mask, float hori_fill, float ray_org_elev,
                       const hz_opts *opts, hz_stats *stats) {
    Timer t; t.start();
    const bool verbose = opts && opts->verbose;
    if (verbose) {   // horizon_comp.cpp:643-645 (the engine named there is Embree)
        printf("--------------------------------------------------------\n");
        printf("Horizon computation with libhorayzon_hip (LBVH on MI355X)\n");
        printf("--------------------------------------------------------\n");
    }
    hz_scene *scene = nullptr;
    hz_stats local;
    memset(&local, 0, sizeof(local));
    int rc = hz_scene_create(vert_grid, dem_dim_0, dem_dim_1, geom_type, vert_simp, num_vert_simp,
                             tri_ind_simp, num_tri_simp, opts ? opts->device : 0, &scene, &local);
    if (rc) return rc;
    if (verbose) {   // horizon_comp.cpp:113-114, :133-135 / :156-158 / :177, :201-203, :227, :664
        printf("DEM dimensions: (%d, %d) \n", dem_dim_0, dem_dim_1);
        printf("Number of vertices: %d \n", dem_dim_0 * dem_dim_1);
        const int nq = (dem_dim_0 - 1) * (dem_dim_1 - 1);
        if (strcmp(geom_type, "triangle") == 0) { printf("Selected geometry type: triangle\n"); printf("Number of triangles: %d \n", nq * 2); }
        else if (strcmp(geom_type, "quad") == 0) { printf("Selected geometry type: quad\n"); printf("Number of quads: %d \n", nq); }
        else printf("Selected geometry type: grid\n");
        if (num_vert_simp >= 3) {
            printf("Add triangles for outer simplified domain\n");
            printf("- number of verties: %d \n", num_vert_simp);
            printf("- number of triangles: %d \n", num_tri_simp);
        }
        printf("BVH build time: %g s\n", local.t_bvh_s);
        printf("Total initialisation time: %g s\n", t.stop());
    }
    rc = hz_horizon_gridded_scene(scene, vec_norm, vec_north, offset_0, offset_1, hori_buffer, dim_in_0,
                                  dim_in_1, azim_num, dist_search, hori_acc, ray_algorithm,
                                  elev_ang_low_lim, mask, hori_fill, ray_org_elev, opts, &local);
    hz_scene_destroy(scene);   // the reference also releases the scene per call, horizon_comp.cpp:813-814
    local.t_total_s = t.stop();
    if (verbose) {   // :818-820
        printf("Total run time: %g s\n", local.t_total_s);
        printf("--------------------------------------------------------\n");
        fflush(stdout);
    }
    if (stats) *stats = local;
    return rc;
}

int hz_horizon_locations_scene(const hz_scene *scene, const float *coords, const float *vec_norm,
                               const float *vec_north, float *hori_buffer, float *hori_dist_buffer, int num_loc,
                               int azim_num, float dist_search, float hori_acc, const char *ray_algorithm,
                               float elev_ang_low_lim, const float *ray_org_elev, int hori_dist_out,
                               const hz_opts *opts, hz_stats *stats) {
    if (!scene) return set_error(HZ_ERR_ARG, "scene is NULL");
    return locations_run(reinterpret_cast<const Scene *>(scene), coords, vec_norm, vec_north, hori_buffer,
                         hori_dist_buffer, num_loc, azim_num, dist_search, hori_acc, ray_algorithm,
                         elev_ang_low_lim, ray_org_elev, hori_dist_out, opts, stats);
}

int hz_horizon_locations(const float *vert_grid, int dem_dim_0, int dem_dim_1, const float *coords,
                         const float *vec_norm, const float *vec_north, float *hori_buffer,
                         float *hori_dist_buffer, int num_loc, int azim_num, float dist_search, float hori_acc,
                         const char *ray_algorithm, const char *geom_type, float elev_ang_low_lim,
                         const float *ray_org_elev, int hori_dist_out, const hz_opts *opts, hz_stats *stats) {
    Timer t; t.start();
    hz_scene *scene = nullptr;
    hz_stats local;
    memset(&local, 0, sizeof(local));
    // no simplified outer mesh for locations: horizon_comp.cpp:848-852
    int rc = hz_scene_create(vert_grid, dem_dim_0, dem_dim_1, geom_type, nullptr, 0, nullptr, 0,
                             opts ? opts->device : 0, &scene, &local);
    if (rc) return rc;
    rc = hz_horizon_locations_scene(scene, coords, vec_norm, vec_north, hori_buffer, hori_dist_buffer, num_loc,
                                    azim_num, dist_search, hori_acc, ray_algorithm, elev_ang_low_lim, ray_org_elev,
                                    hori_dist_out, opts, &local);
    hz_scene_destroy(scene);
    local.t_total_s = t.stop();
    if (stats) *stats = local;
    return rc;
}

int hz_horizon_tables(int azim_num, float hori_acc, float elev_ang_low_lim, float *azim_sin,
                      float *azim_cos, int elev_cap, float *elev_ang, float *elev_sin, float *elev_cos,
                      int *elev_num) {
    if (azim_num <= 0 || !(hori_acc > 0.0f)) return set_error(HZ_ERR_ARG, "azim_num and hori_acc must be positive");
    HostTables t;
    build_tables(azim_num, hori_acc, elev_ang_low_lim, t);
    if (elev_num) *elev_num = t.elev_num;
    if (azim_sin) memcpy(azim_sin, t.azim_sin.data(), sizeof(float) * (size_t)azim_num);
    if (azim_cos) memcpy(azim_cos, t.azim_cos.data(), sizeof(float) * (size_t)azim_num);
    if (elev_ang && elev_sin && elev_cos && t.elev_num > 0 && t.elev_num <= elev_cap) {
        memcpy(elev_ang, t.elev_ang.data(), sizeof(float) * (size_t)t.elev_num);
        memcpy(elev_sin, t.elev_sin.data(), sizeof(float) * (size_t)t.elev_num);
        memcpy(elev_cos, t.elev_cos.data(), sizeof(float) * (size_t)t.elev_num);
    }
    return HZ_OK;
}

static int topo_api(int kind, const float *azim, const float *hori, const float *vec_tilt, int len_0, int len_1,
                    int len_2, float *out, int device) {
    if (!azim || !hori || !out || (kind != 2 && !vec_tilt)) return set_error(HZ_ERR_ARG, "NULL argument");
    if (len_0 <= 0 || len_1 <= 0 || len_2 < 2) return set_error(HZ_ERR_ARG, "Inconsistent/incorrect shapes of input arrays");
    int rc = select_device(device);
    if (rc) return rc;
    hipStream_t st = nullptr;
    const size_t ncell = (size_t)len_0 * len_1;
    DevIn<float> d_azim, d_hori, d_tilt;
    DevOut<float> d_out;
    if ((rc = d_azim.bind(azim, (size_t)len_2, st))) return rc;
    if ((rc = d_hori.bind(hori, ncell * (size_t)len_2, st))) return rc;
    if (kind != 2) if ((rc = d_tilt.bind(vec_tilt, ncell * 3, st))) return rc;
    if ((rc = d_out.bind(out, ncell))) return rc;
    if ((rc = topo_launch(kind, d_azim.dev, d_hori.dev, d_tilt.dev, len_0, len_1, len_2, d_out.dev, st))) return rc;
    if ((rc = d_out.finish(st))) return rc;
    HZ_HIP(hipStreamSynchronize(st));
    return HZ_OK;
}

int hz_sky_view_factor(const float *azim, const float *hori, const float *vec_tilt, int len_0, int len_1,
                       int len_2, float *svf, int device) {
    return topo_api(0, azim, hori, vec_tilt, len_0, len_1, len_2, svf, device);
}
int hz_visible_sky_fraction(const float *azim, const float *hori, const float *vec_tilt, int len_0, int len_1,
                            int len_2, float *vsf, int device) {
    return topo_api(1, azim, hori, vec_tilt, len_0, len_1, len_2, vsf, device);
}
int hz_topographic_openness(const float *azim, const float *hori, int len_0, int len_1, int len_2, float *top,
                            int device) {
    return topo_api(2, azim, hori, nullptr, len_0, len_1, len_2, top, device);
}

// ---------------------------------------------------------------------------------------
// test hooks for the build primitives (hz_sort.hip)
// ---------------------------------------------------------------------------------------
int hz_debug_sort_pairs(uint32_t *keys, uint32_t *vals, size_t n, int device) {
    if (!keys || !vals) return set_error(HZ_ERR_ARG, "NULL argument");
    int rc = select_device(device);
    if (rc) return rc;
    hipStream_t st = nullptr;
    void *dk = nullptr, *dv = nullptr, *dk2 = nullptr, *dv2 = nullptr, *tmp = nullptr;
    const size_t bytes = (n ? n : 1) * 4;
    HZ_HIP(hipMalloc(&dk, bytes)); HZ_HIP(hipMalloc(&dv, bytes)); HZ_HIP(hipMalloc(&dk2, bytes)); HZ_HIP(hipMalloc(&dv2, bytes));
    HZ_HIP(hipMalloc(&tmp, sort_temp_elems(n) * 4 + 16));
    HZ_HIP(hipMemcpy(dk, keys, n * 4, hipMemcpyHostToDevice));
    HZ_HIP(hipMemcpy(dv, vals, n * 4, hipMemcpyHostToDevice));
    rc = radix_sort_pairs_u32((uint32_t *)dk, (uint32_t *)dv, (uint32_t *)dk2, (uint32_t *)dv2, n, (uint32_t *)tmp, st);
    if (!rc) {
        HZ_HIP(hipStreamSynchronize(st));
        HZ_HIP(hipMemcpy(keys, dk, n * 4, hipMemcpyDeviceToHost));
        HZ_HIP(hipMemcpy(vals, dv, n * 4, hipMemcpyDeviceToHost));
    }
    (void)hipFree(dk); (void)hipFree(dv); (void)hipFree(dk2); (void)hipFree(dv2); (void)hipFree(tmp);
    return rc;
}

int hz_debug_exclusive_scan(const uint32_t *in, uint32_t *out, size_t n, int device) {
    if (!in || !out) return set_error(HZ_ERR_ARG, "NULL argument");
    int rc = select_device(device);
    if (rc) return rc;
    hipStream_t st = nullptr;
    void *di = nullptr, *dout = nullptr, *tmp = nullptr;
    const size_t bytes = (n ? n : 1) * 4;
    HZ_HIP(hipMalloc(&di, bytes)); HZ_HIP(hipMalloc(&dout, bytes)); HZ_HIP(hipMalloc(&tmp, scan_temp_elems(n) * 4 + 16));
    HZ_HIP(hipMemcpy(di, in, n * 4, hipMemcpyHostToDevice));
    rc = exclusive_scan_u32((const uint32_t *)di, (uint32_t *)dout, n, (uint32_t *)tmp, st);
    if (!rc) {
        HZ_HIP(hipStreamSynchronize(st));
        HZ_HIP(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    }
    (void)hipFree(di); (void)hipFree(dout); (void)hipFree(tmp);
    return rc;
}

int hz_debug_valu_peak(int device, int packed, int waves_per_simd, double *winst_per_s_per_simd,
                       double *clock_ghz, int *simds) {
    int rc = select_device(device);
    if (rc) return rc;
    return bench_valu_peak(packed, waves_per_simd, winst_per_s_per_simd, clock_ghz, simds);
}

int hz_debug_set(const char *key, int value) {
    if (!key) return hz::set_error(HZ_ERR_ARG, "hz_debug_set: null key");
    if (!strcmp(key, "shadow_fast_cap")) hz::g_shadow_fast_cap.store(value < 0 ? HZ_SHADOW_FAST_CAP_DEFAULT : value, std::memory_order_relaxed);
    else if (!strcmp(key, "topo_wide")) hz::g_topo_wide.store(value != 0, std::memory_order_relaxed);
    else return hz::set_error(HZ_ERR_ARG, "hz_debug_set: unknown key '%s'", key);
    return HZ_OK;
}

int hz_debug_inst_rate(int device, int op, double *cycles_per_inst) {
    int rc = select_device(device);
    if (rc) return rc;
    return bench_inst_rate(op, cycles_per_inst);
}

int hz_debug_copy_peak(int device, size_t bytes, double *gbs) {
    int rc = select_device(device);
    if (rc) return rc;
    return bench_copy_peak(bytes, gbs);
}

// ---------------------------------------------------------------------------------------
// slope and input preparation (hz_prep.hip)
// ---------------------------------------------------------------------------------------
static int slope_api(int which, const float *x, const float *y, const float *z, int len_0, int len_1,
                     const float *rot_mat, int output_rot, float *vec_tilt, int device) {
    if (!x || !y || !z || !vec_tilt) return set_error(HZ_ERR_ARG, "NULL argument");
    if (len_0 <= 0 || len_1 <= 0) return set_error(HZ_ERR_ARG, "Inconsistent shapes / number of dimensions of input arrays");
    if (which == 1 && output_rot && !rot_mat) return set_error(HZ_ERR_ARG, "'rot_mat' must be provided for 'output_rot = True'");
    int rc = select_device(device);
    if (rc) return rc;
    hipStream_t st = nullptr;
    const size_t n = (size_t)len_0 * len_1;
    DevIn<float> dx, dy, dz, dr;
    DevOut<float> dt;
    if ((rc = dx.bind(x, n, st))) return rc;
    if ((rc = dy.bind(y, n, st))) return rc;
    if ((rc = dz.bind(z, n, st))) return rc;
    if ((rc = dr.bind(rot_mat, rot_mat ? n * 9 : 0, st))) return rc;
    if ((rc = dt.bind(vec_tilt, n * 3))) return rc;
    if ((rc = prep_slope(which, dx.dev, dy.dev, dz.dev, len_0, len_1, dr.dev, output_rot, dt.dev, st))) return rc;
    if ((rc = dt.finish(st))) return rc;
    HZ_HIP(hipStreamSynchronize(st));
    return HZ_OK;
}

int hz_slope_plane_meth(const float *x, const float *y, const float *z, int len_0, int len_1,
                        const float *rot_mat, int output_rot, float *vec_tilt, int device) {
    return slope_api(0, x, y, z, len_0, len_1, rot_mat, output_rot, vec_tilt, device);
}
int hz_slope_vector_meth(const float *x, const float *y, const float *z, int len_0, int len_1,
                         const float *rot_mat, int output_rot, float *vec_tilt, int device) {
    return slope_api(1, x, y, z, len_0, len_1, rot_mat, output_rot, vec_tilt, device);
}

static int check_ellps(int ellps) {
    if (ellps < 0 || ellps > 2) return set_error(HZ_ERR_ARG, "Unknown value for 'ellps'");
    return HZ_OK;
}

int hz_lonlat2ecef(const double *lon, const double *lat, const float *h, size_t n, int ellps, double *x_ecef,
                   double *y_ecef, double *z_ecef, int device) {
    if (!lon || !lat || !h || !x_ecef || !y_ecef || !z_ecef) return set_error(HZ_ERR_ARG, "NULL argument");
    int rc = check_ellps(ellps);
    if (rc) return rc;
    if ((rc = select_device(device))) return rc;
    hipStream_t st = nullptr;
    DevIn<double> dlon, dlat; DevIn<float> dh; DevOut<double> ox, oy, oz;
    if ((rc = dlon.bind(lon, n, st)) || (rc = dlat.bind(lat, n, st)) || (rc = dh.bind(h, n, st))) return rc;
    if ((rc = ox.bind(x_ecef, n)) || (rc = oy.bind(y_ecef, n)) || (rc = oz.bind(z_ecef, n))) return rc;
    if ((rc = prep_lonlat2ecef(ellps, dlon.dev, dlat.dev, dh.dev, n, ox.dev, oy.dev, oz.dev, st))) return rc;
    if ((rc = ox.finish(st)) || (rc = oy.finish(st)) || (rc = oz.finish(st))) return rc;
    HZ_HIP(hipStreamSynchronize(st));
    return HZ_OK;
}

int hz_wgs2swiss(const double *lon, const double *lat, const float *h_wgs, size_t n, double *e, double *no, float *h_ch,
                 int device) {
    if (!lon || !lat || !h_wgs || !e || !no || !h_ch) return set_error(HZ_ERR_ARG, "NULL argument");
    int rc = select_device(device);
    if (rc) return rc;
    hipStream_t st = nullptr;
    DevIn<double> da, db; DevIn<float> dh; DevOut<double> oa, ob; DevOut<float> oh;
    if ((rc = da.bind(lon, n, st)) || (rc = db.bind(lat, n, st)) || (rc = dh.bind(h_wgs, n, st))) return rc;
    if ((rc = oa.bind(e, n)) || (rc = ob.bind(no, n)) || (rc = oh.bind(h_ch, n))) return rc;
    if ((rc = prep_wgs2swiss(da.dev, db.dev, dh.dev, n, oa.dev, ob.dev, oh.dev, st))) return rc;
    if ((rc = oa.finish(st)) || (rc = ob.finish(st)) || (rc = oh.finish(st))) return rc;
    HZ_HIP(hipStreamSynchronize(st));
    return HZ_OK;
}

int hz_swiss2wgs(const double *e, const double *no, const float *h_ch, size_t n, double *lon, double *lat, float *h_wgs,
                 int device) {
    if (!e || !no || !h_ch || !lon || !lat || !h_wgs) return set_error(HZ_ERR_ARG, "NULL argument");
    int rc = select_device(device);
    if (rc) return rc;
    hipStream_t st = nullptr;
    DevIn<double> da, db; DevIn<float> dh; DevOut<double> oa, ob; DevOut<float> oh;
    if ((rc = da.bind(e, n, st)) || (rc = db.bind(no, n, st)) || (rc = dh.bind(h_ch, n, st))) return rc;
    if ((rc = oa.bind(lon, n)) || (rc = ob.bind(lat, n)) || (rc = oh.bind(h_wgs, n))) return rc;
    if ((rc = prep_swiss2wgs(da.dev, db.dev, dh.dev, n, oa.dev, ob.dev, oh.dev, st))) return rc;
    if ((rc = oa.finish(st)) || (rc = ob.finish(st)) || (rc = oh.finish(st))) return rc;
    HZ_HIP(hipStreamSynchronize(st));
    return HZ_OK;
}

int hz_ecef2enu(const double *x_ecef, const double *y_ecef, const double *z_ecef, size_t n, double lon_or,
                double lat_or, int ellps, float *x_enu, float *y_enu, float *z_enu, int device) {
    if (!x_ecef || !y_ecef || !z_ecef || !x_enu || !y_enu || !z_enu) return set_error(HZ_ERR_ARG, "NULL argument");
    if (lon_or < -180.0 || lon_or > 180.0) return set_error(HZ_ERR_ARG, "Value for 'lon_or' is outside of valid range");
    if (lat_or < -90.0 || lat_or > 90.0) return set_error(HZ_ERR_ARG, "Value for 'lat_or' is outside of valid range");
    int rc = check_ellps(ellps);
    if (rc) return rc;
    if ((rc = select_device(device))) return rc;
    hipStream_t st = nullptr;
    DevIn<double> dx, dy, dz; DevOut<float> ox, oy, oz;
    if ((rc = dx.bind(x_ecef, n, st)) || (rc = dy.bind(y_ecef, n, st)) || (rc = dz.bind(z_ecef, n, st))) return rc;
    if ((rc = ox.bind(x_enu, n)) || (rc = oy.bind(y_enu, n)) || (rc = oz.bind(z_enu, n))) return rc;
    if ((rc = prep_ecef2enu(ellps, lon_or, lat_or, dx.dev, dy.dev, dz.dev, n, ox.dev, oy.dev, oz.dev, st))) return rc;
    if ((rc = ox.finish(st)) || (rc = oy.finish(st)) || (rc = oz.finish(st))) return rc;
    HZ_HIP(hipStreamSynchronize(st));
    return HZ_OK;
}

int hz_ecef2enu_vector(const float *vec_ecef, size_t n, double lon_or, double lat_or, int ellps, float *vec_enu,
                       int device) {
    if (!vec_ecef || !vec_enu) return set_error(HZ_ERR_ARG, "NULL argument");
    int rc = check_ellps(ellps);
    if (rc) return rc;
    if ((rc = select_device(device))) return rc;
    hipStream_t st = nullptr;
    DevIn<float> dv; DevOut<float> dout;
    if ((rc = dv.bind(vec_ecef, n * 3, st)) || (rc = dout.bind(vec_enu, n * 3))) return rc;
    if ((rc = prep_ecef2enu_vector(ellps, lon_or, lat_or, dv.dev, n, dout.dev, st))) return rc;
    if ((rc = dout.finish(st))) return rc;
    HZ_HIP(hipStreamSynchronize(st));
    return HZ_OK;
}

int hz_surf_norm(const double *lon, const double *lat, size_t n, float *vec_norm_ecef, int device) {
    if (!lon || !lat || !vec_norm_ecef) return set_error(HZ_ERR_ARG, "NULL argument");
    int rc = select_device(device);
    if (rc) return rc;
    hipStream_t st = nullptr;
    DevIn<double> dlon, dlat; DevOut<float> dout;
    if ((rc = dlon.bind(lon, n, st)) || (rc = dlat.bind(lat, n, st)) || (rc = dout.bind(vec_norm_ecef, n * 3))) return rc;
    if ((rc = prep_surf_norm(dlon.dev, dlat.dev, n, dout.dev, st))) return rc;
    if ((rc = dout.finish(st))) return rc;
    HZ_HIP(hipStreamSynchronize(st));
    return HZ_OK;
}

int hz_north_dir(const double *x_ecef, const double *y_ecef, const double *z_ecef, const float *vec_norm_ecef,
                 size_t n, int ellps, float *vec_north_ecef, int device) {
    if (!x_ecef || !y_ecef || !z_ecef || !vec_norm_ecef || !vec_north_ecef) return set_error(HZ_ERR_ARG, "NULL argument");
    int rc = check_ellps(ellps);
    if (rc) return rc;
    if ((rc = select_device(device))) return rc;
    hipStream_t st = nullptr;
    DevIn<double> dx, dy, dz; DevIn<float> dn; DevOut<float> dout;
    if ((rc = dx.bind(x_ecef, n, st)) || (rc = dy.bind(y_ecef, n, st)) || (rc = dz.bind(z_ecef, n, st))) return rc;
    if ((rc = dn.bind(vec_norm_ecef, n * 3, st)) || (rc = dout.bind(vec_north_ecef, n * 3))) return rc;
    if ((rc = prep_north_dir(ellps, dx.dev, dy.dev, dz.dev, dn.dev, n, dout.dev, st))) return rc;
    if ((rc = dout.finish(st))) return rc;
    HZ_HIP(hipStreamSynchronize(st));
    return HZ_OK;
}

size_t hz_vert_grid_len(size_t num_vertices) {
    // pad_buffer (auxiliary.py:128-131): 16 extra elements, plus what makes the byte size a multiple of 16
    const size_t n3 = 3 * num_vertices;
    size_t add = 16;
    if ((n3 * 4) % 16 != 0) add += (16 - (n3 * 4) % 16) / 4;
    return n3 + add;
}

int hz_pack_vertices(const float *x, const float *y, const float *z, size_t num_vertices, float *vert_grid,
                     size_t vert_grid_len, int device) {
    if (!x || !y || !z || !vert_grid) return set_error(HZ_ERR_ARG, "NULL argument");
    if (vert_grid_len < hz_vert_grid_len(num_vertices))
        return set_error(HZ_ERR_ARG, "vert_grid is shorter than hz_vert_grid_len(num_vertices)");
    int rc = select_device(device);
    if (rc) return rc;
    hipStream_t st = nullptr;
    DevIn<float> dx, dy, dz; DevOut<float> dout;
    if ((rc = dx.bind(x, num_vertices, st)) || (rc = dy.bind(y, num_vertices, st)) || (rc = dz.bind(z, num_vertices, st))) return rc;
    if ((rc = dout.bind(vert_grid, vert_grid_len))) return rc;
    if ((rc = prep_pack_vertices(dx.dev, dy.dev, dz.dev, num_vertices, vert_grid_len, dout.dev, st))) return rc;
    if ((rc = dout.finish(st))) return rc;
    HZ_HIP(hipStreamSynchronize(st));
    return HZ_OK;
}

// ---------------------------------------------------------------------------------------
// Terrain (shadow_comp.h:4-39)
// ---------------------------------------------------------------------------------------

int hz_terrain_create(int device, hz_terrain **terrain) {
    if (!terrain) return set_error(HZ_ERR_ARG, "terrain is NULL");
    int rc = select_device(device);
    if (rc) return rc;
    Terrain *t = new Terrain();
    t->device = device;
    *terrain = reinterpret_cast<hz_terrain *>(t);
    return HZ_OK;
}

static int terrain_init_common(Terrain *t, int offset_0, int offset_1, const float *vec_tilt,
                               const float *vec_norm, int dim_in_0, int dim_in_1, const float *surf_enl_fac,
                               const float *elevation, const uint8_t *mask, float sw_dir_cor_fill,
                               float ang_max, int refrac_cor) {
    const Scene *sc = t->scene;
    if (!vec_tilt || !vec_norm || !surf_enl_fac || !elevation || !mask) return set_error(HZ_ERR_ARG, "NULL input array");
    if (dim_in_0 <= 0 || dim_in_1 <= 0 || offset_0 < 0 || offset_1 < 0 ||
        offset_0 + dim_in_0 > sc->hdr.d0 || offset_1 + dim_in_1 > sc->hdr.d1)
        return set_error(HZ_ERR_ARG, "inconsistency between input arguments 'dem_dim_0', 'dem_dim_1', 'offset_0', 'offset_1' and 'vec_norm'");
    if (ang_max < 85.0f || ang_max > 89.99f) return set_error(HZ_ERR_ARG, "'ang_max' must be in the range [85.0, 89.99]");
    std::lock_guard<std::mutex> run_lock(sc->run_mu);
    hipStream_t st = sc->stream;
    const size_t nc = (size_t)dim_in_0 * dim_in_1;
    int rc;
    if ((rc = persist(vec_tilt, nc * 12, st, &t->tilt, &t->own_tilt))) return rc;
    if ((rc = persist(vec_norm, nc * 12, st, &t->norm, &t->own_norm))) return rc;
    if ((rc = persist(surf_enl_fac, nc * 4, st, &t->enl, &t->own_enl))) return rc;
    if ((rc = persist(elevation, nc * 4, st, &t->elev, &t->own_elev))) return rc;
    if ((rc = persist(mask, nc, st, &t->mask, &t->own_mask))) return rc;
    if (!t->counters) HZ_HIP(hipMalloc((void **)&t->counters, 16 * sizeof(unsigned long long)));
    if (refrac_cor) {
        // what the refraction formula needs from the cell's elevation alone -- temperature, pressure (a powf), two float64
        // divisions -- is the same for every sun position: formed once here (shadow_comp.cpp:438-441, :151-157)
        HZ_HIP(hipMalloc((void **)&t->refrac_fac, nc * sizeof(double)));
        if ((rc = shadow_refrac_factor((const float *)t->elev, nc, t->refrac_fac, st))) return rc;
    }
    HZ_HIP(hipStreamSynchronize(st));
    t->offset_0 = offset_0; t->offset_1 = offset_1; t->dim_in_0 = dim_in_0; t->dim_in_1 = dim_in_1;
    t->fill = sw_dir_cor_fill; t->ang_max = ang_max; t->refrac = refrac_cor ? 1 : 0;
    t->stream = st;
    t->initialised = true;
    return HZ_OK;
}

int hz_terrain_initialise(hz_terrain *terrain, const float *vert_grid, int dem_dim_0, int dem_dim_1,
                          int offset_0, int offset_1, const float *vec_tilt, const float *vec_norm,
                          int dim_in_0, int dim_in_1, const float *surf_enl_fac, const float *elevation,
                          const uint8_t *mask, const char *geom_type, float sw_dir_cor_fill, float ang_max,
                          int refrac_cor, hz_stats *stats) {
    if (!terrain) return set_error(HZ_ERR_ARG, "terrain is NULL");
    Terrain *t = reinterpret_cast<Terrain *>(terrain);
    terrain_release_arrays(t);
    HZ_HIP(hipSetDevice(t->device));
    hz_scene *scene = nullptr;
    // no simplified outer TIN in the shadow scene: shadow_comp.cpp:198-298
    int rc = hz_scene_create(vert_grid, dem_dim_0, dem_dim_1, geom_type, nullptr, 0, nullptr, 0, t->device,
                             &scene, stats);
    if (rc) return rc;
    t->scene = reinterpret_cast<Scene *>(scene);
    t->owns_scene = true;
    rc = terrain_init_common(t, offset_0, offset_1, vec_tilt, vec_norm, dim_in_0, dim_in_1, surf_enl_fac,
                             elevation, mask, sw_dir_cor_fill, ang_max, refrac_cor);
    if (rc) terrain_release_arrays(t);
    return rc;
}

int hz_terrain_initialise_scene(hz_terrain *terrain, const hz_scene *scene, int offset_0, int offset_1,
                                const float *vec_tilt, const float *vec_norm, int dim_in_0, int dim_in_1,
                                const float *surf_enl_fac, const float *elevation, const uint8_t *mask,
                                float sw_dir_cor_fill, float ang_max, int refrac_cor) {
    if (!terrain || !scene) return set_error(HZ_ERR_ARG, "terrain / scene is NULL");
    Terrain *t = reinterpret_cast<Terrain *>(terrain);
    terrain_release_arrays(t);
    t->scene = const_cast<Scene *>(reinterpret_cast<const Scene *>(scene));
    t->owns_scene = false;
    // the terrain lives on the scene's GPU: its per-cell arrays are allocated there and its kernels run on the
    // scene's stream (a Terrain created for another ordinal follows the scene)
    t->device = t->scene->device;
    HZ_HIP(hipSetDevice(t->device));
    int rc = terrain_init_common(t, offset_0, offset_1, vec_tilt, vec_norm, dim_in_0, dim_in_1, surf_enl_fac,
                                 elevation, mask, sw_dir_cor_fill, ang_max, refrac_cor);
    if (rc) terrain_release_arrays(t);
    return rc;
}

static int terrain_run(Terrain *t, const float *sun_positions, int num_sun, int which, uint8_t *out_u8,
                       float *out_f32, hz_stats *stats) {
    if (!t || !t->initialised) return set_error(HZ_ERR_ARG, "Terrain is not initialised");
    if (!sun_positions || num_sun <= 0) return set_error(HZ_ERR_ARG, "array 'sun_position' has incorrect shape");
    if ((which == 0 && !out_u8) || (which == 1 && !out_f32)) return set_error(HZ_ERR_ARG, "output buffer is NULL");
    HZ_HIP(hipSetDevice(t->device));
    std::lock_guard<std::mutex> run_lock(t->scene->run_mu);
    hipStream_t st = t->stream;
    Timer t_total; t_total.start();
    const size_t nc = (size_t)t->dim_in_0 * t->dim_in_1;
    std::vector<float> sun((size_t)num_sun * 3);
    if (is_device_ptr(sun_positions)) HZ_HIP(hipMemcpy(sun.data(), sun_positions, sun.size() * 4, hipMemcpyDeviceToHost));
    else memcpy(sun.data(), sun_positions, sun.size() * 4);
    DevOut<uint8_t> d_u8; DevOut<float> d_f32;
    int rc;
    if (which == 0) { if ((rc = d_u8.bind(out_u8, nc * (size_t)num_sun))) return rc; }
    else { if ((rc = d_f32.bind(out_f32, nc * (size_t)num_sun))) return rc; }
    ShadowArgs a;
    a.vec_tilt = (const float *)t->tilt; a.vec_norm = (const float *)t->norm;
    a.surf_enl_fac = (const float *)t->enl; a.elevation = (const float *)t->elev; a.mask = (const uint8_t *)t->mask;
    a.offset_0 = t->offset_0; a.offset_1 = t->offset_1; a.dim_in_0 = t->dim_in_0; a.dim_in_1 = t->dim_in_1;
    a.sw_dir_cor_fill = t->fill;
    a.dot_prod_min = cosf(deg2rad_f(t->ang_max));            // shadow_comp.cpp:498
    a.refrac_cor = t->refrac; a.refrac_fac = t->refrac_fac; a.which = which; a.top_nodes = -1; a.counters = t->counters;
    a.count_work = t->count_work;
    float ms = 0.0f;
    unsigned long long cnt[16];
    {
        // the sun positions go to the device once; one launch computes up to 32768 of them (grid.y)
        if (t->sun_cap < sun.size()) {
            if (t->sun_dev) (void)hipFree(t->sun_dev);
            t->sun_dev = nullptr; t->sun_cap = 0;
            const size_t cap = std::max<size_t>(sun.size(), 3 * 256);
            HZ_HIP(hipMalloc((void **)&t->sun_dev, cap * sizeof(float)));
            t->sun_cap = cap;
        }
        HZ_HIP(hipMemcpyAsync(t->sun_dev, sun.data(), sun.size() * sizeof(float), hipMemcpyHostToDevice, st));
        HZ_HIP(hipMemsetAsync(t->counters, 0, 16 * sizeof(unsigned long long), st));
        hipEvent_t e0, e1;
        HZ_HIP(hipEventCreate(&e0)); HZ_HIP(hipEventCreate(&e1));
        HZ_HIP(hipEventRecord(e0, st));
        for (int s0 = 0; s0 < num_sun; s0 += 32768) {
            a.suns = t->sun_dev + 3 * (size_t)s0;
            a.num_sun = std::min(32768, num_sun - s0);
            a.out_u8 = d_u8.dev ? d_u8.dev + nc * (size_t)s0 : nullptr;
            a.out_f32 = d_f32.dev ? d_f32.dev + nc * (size_t)s0 : nullptr;
            if ((rc = shadow_launch(t->scene, a, st))) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
        }
        HZ_HIP(hipEventRecord(e1, st));
        HZ_HIP(hipEventSynchronize(e1));
        HZ_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        HZ_HIP(hipMemcpyAsync(cnt, t->counters, sizeof(cnt), hipMemcpyDeviceToHost, st));
        HZ_HIP(hipStreamSynchronize(st));
    }
    Timer t_d2h; t_d2h.start();
    if ((rc = d_u8.finish(st))) return rc;
    if ((rc = d_f32.finish(st))) return rc;
    HZ_HIP(hipStreamSynchronize(st));
    if (stats) {
        stats->num_rays += cnt[0];
        stats->nodes_visited += cnt[1]; stats->tris_tested += cnt[2];
        stats->wave_node_iters += cnt[3]; stats->wave_leaf_iters += cnt[4];
        stats->t_kernel_s += (double)ms * 1e-3;
        stats->t_d2h_s += t_d2h.stop();
        stats->t_total_s += t_total.stop();
        stats->bvh_height = t->scene->hdr.height; stats->scene_bytes = t->scene->hdr.total_bytes;
    }
    return HZ_OK;
}

int hz_terrain_count_work(hz_terrain *terrain, int on) {
    if (!terrain) return set_error(HZ_ERR_ARG, "terrain is NULL");
    reinterpret_cast<Terrain *>(terrain)->count_work = on ? 1 : 0;
    return HZ_OK;
}

int hz_terrain_shadow(hz_terrain *terrain, const float *sun_position, uint8_t *shadow_buffer, hz_stats *stats) {
    return terrain_run(reinterpret_cast<Terrain *>(terrain), sun_position, 1, 0, shadow_buffer, nullptr, stats);
}
int hz_terrain_sw_dir_cor(hz_terrain *terrain, const float *sun_position, float *sw_dir_cor_buffer, hz_stats *stats) {
    return terrain_run(reinterpret_cast<Terrain *>(terrain), sun_position, 1, 1, nullptr, sw_dir_cor_buffer, stats);
}
int hz_terrain_shadow_batch(hz_terrain *terrain, const float *sun_positions, int num_sun,
                            uint8_t *shadow_buffers, hz_stats *stats) {
    return terrain_run(reinterpret_cast<Terrain *>(terrain), sun_positions, num_sun, 0, shadow_buffers, nullptr, stats);
}
int hz_terrain_sw_dir_cor_batch(hz_terrain *terrain, const float *sun_positions, int num_sun,
                                float *sw_dir_cor_buffers, hz_stats *stats) {
    return terrain_run(reinterpret_cast<Terrain *>(terrain), sun_positions, num_sun, 1, nullptr, sw_dir_cor_buffers, stats);
}

int hz_terrain_destroy(hz_terrain *terrain) {
    Terrain *t = reinterpret_cast<Terrain *>(terrain);
    if (!t) return HZ_OK;
    (void)hipSetDevice(t->device);
    terrain_release_arrays(t);
    if (t->counters) (void)hipFree(t->counters);
    delete t;
    return HZ_OK;
}

}  // extern "C"
