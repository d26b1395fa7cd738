// dspi_capi.cpp — the C-ABI of libdspi_mi355x (include/dspi.h): context, device memory, launches.
//
// The context owns, per GPU: one state array, one delay-line array and one leveller ring for all
// streams (laid out per 64-stream workgroup, see dspi_image.h / DESIGN.md), plus a table of
// parameter images.  Streams reference images; a parameter call addressed to a single stream
// that shares its image clones the image first (copy-on-write), so "same preset on every
// stream" stays one broadcast image and per-stream presets still work — each distinct image is
// one launch over the workgroups (and lane masks) that use it.
#include <hip/hip_runtime.h>

#include <string.h>

#include <memory>
#include <new>
#include <cstdlib>
#include <string>
#include <map>
#include <unordered_map>
#include <type_traits>
#include <vector>
#include <thread>
#include <xmmintrin.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include "../../include/dspi_detmath.h"

#include "../../include/dspi.h"
#include "dspi_image.h"
#include "dspi_kernels.h"
#include "dspi_params.h"

using namespace dspi;

struct dspi_ctx {
    int flavor = 1;
    bool fma = false;              // DSPI_FLOAT_CONTRACT_FMA: host design and kernels use the firmware build's fused multiply-adds
    uint32_t n_streams = 0, n_wg = 0;
    int device = DSPI_DEVICE_NONE;
    StateMap sm{};
    std::vector<std::unique_ptr<Params>> images;
    std::vector<uint32_t> image_refs;
    std::vector<int32_t> stream_image;
    bool assignment_dirty = true;
    bool merge_hint = false;       // a broadcast call ran while several images were live: equal images fold back into one (merge_images)
    // per image, four work lists (dspi_image.h:WgItem): 0 = all streams of the image (state ops; the Q28 chain launch),
    // 1 = lanes whose two streams both belong (float: packed kernel), 2/3 = lanes where only stream 0 / 1 belongs
    std::vector<std::vector<WgItem>> image_items[4];
    std::vector<uint32_t> image_item_offset[4];
    // chain launches: the same work lists concatenated over images, grouped by what the kernels are specialised on
    // (float: leveller on/off), so a dspi_process is a handful of launches however many presets are in play
    std::vector<WgItem> launch_items[2][9];      // [leveller off / on][list]; lists 5, 6: the latency layout (dspi_chain_skew.inc) without / with output rows;
    uint32_t launch_item_offset[2][9] = {};      // 7, 8: the same with paired presets (workgroups whose stream slots hold different images of one structure)
    WgItem *d_litems = nullptr; size_t d_litems_cap = 0;
    uint32_t *d_stream_image = nullptr; size_t d_stream_image_cap = 0;   // image index per stream (per-lane parameter kernel)
    bool launch_dirty = true;
    uint32_t spdif_pos = 0;      // DSPI_OUT_SPDIF: block position of the next call's first frame
    bool populated = false;      // DSPI_BOOT_POPULATED_FLASH: the streams are devices whose flash already holds a preset directory
    bool audio_started = false;  // a dspi_process has run: the devices are no longer booting (dspi_load_flash_dump)
    bool no_direct = false;      // DSPI_NO_DIRECT (development / tests, read once at dspi_create): the staged path for small host calls too
    // the leveller's alpha^count on the device is step 1 + a table that was generated for the firmware's 18 alphas (include/dspi_detmath.h); the
    // alphas this context has actually built into images, and the block length they were last checked with against the exact form
    std::vector<uint32_t> lv_alphas; uint32_t lv_checked_count = 0; size_t lv_checked_n = 0;
    uint32_t direct_spin_us = 0; // DSPI_DIRECT_SPIN_US (read once at dspi_create): how long a direct call polls its stream before the blocking wait; 0 = the call's own audio time (>= 300 us)
    uint64_t direct_stats[5] = {0, 0, 0, 0, 0};      // dspi_debug_direct_stats: calls, calls that fell back to the blocking wait, max enqueue ns, max wait ns, last spin budget ns
    // device
    hipStream_t hs = nullptr;
    // host-buffer dspi_process: H2D, kernels and D2H of consecutive row chunks overlap on three streams (created on first use)
    hipStream_t hs_in = nullptr, hs_out = nullptr;
    std::vector<hipEvent_t> pipe_events;
    uint32_t *d_state = nullptr, *d_dlines = nullptr, *d_ring = nullptr;
    uint32_t *d_xwords = nullptr; size_t d_xwords_cap = 0;      // exchange area of the packed kernel's copy wave (stream-major output)
    DevImage *d_images = nullptr;
    std::vector<uint32_t> image_flags;             // DevImage::flags of each uploaded image (kernel variant selection)
    // float flavour, per-lane VALUES: a row whose streams carry several presets of one structure runs the packed kernel with
    // its numbers in a value tile (dspi_image.h); ImageSig = what has to agree for that
    struct ImageSig {
        uint32_t flags, ch_bypassed, out_enabled, out_mute, fs_hz, mute_transition, mix_nz, i2s_pairs;
        int32_t delay[kMaxOut];
        uint8_t kinds[kPvBandSlots];
    };
    struct BandHash { uint64_t a, b; };            // two independent 64-bit hashes of an image's band coefficient words: rows whose
    std::vector<BandHash> image_bands;             // images agree in both run the shared band loops (row_pv = 2), the rest of the
                                                   // numbers per lane
    std::vector<ImageSig> image_sig;
    std::vector<uint8_t> row_pv;                   // [n_wg] the row is a per-lane-value row
    std::vector<uint8_t> image_touched;            // images uploaded since the tiles were last built
    float *d_vals = nullptr; size_t d_vals_cap = 0;
    uint32_t *d_pv_rows = nullptr; size_t d_pv_rows_cap = 0;
    size_t d_images_cap = 0;
    WgItem *d_items = nullptr;
    size_t d_items_cap = 0;
    // staging buffers for host-memory dspi_process
    void *d_in = nullptr; size_t d_in_cap = 0;
    int32_t *d_pairs = nullptr; size_t d_pairs_cap = 0;
    int32_t *d_sub = nullptr; size_t d_sub_cap = 0;
    uint16_t *d_peaks = nullptr; size_t d_peaks_cap = 0;
    uint16_t *d_clip = nullptr; size_t d_clip_cap = 0;          // DSPI_OUT_CLIP_FLAGS on host buffers
    // small calls on host buffers (one packet per call, the firmware's own rhythm): a pinned host area the kernels read and write directly
    char *h_direct = nullptr; char *d_direct = nullptr; size_t direct_cap = 0;
    // ... and a completion word of its own next to that area (round 6): the stream writes the call's sequence number there once the launches
    // have ended (hipStreamWriteValue32) and the host polls MEMORY instead of calling into the runtime (hipStreamQuery) thousands of times
    uint32_t *h_done = nullptr; uint32_t *d_done = nullptr; uint32_t direct_seq = 0; int direct_flag = -1;      // -1: untried, 0: not available / DSPI_DIRECT_POLL=query, 1: in use
    int32_t *d_spdif_words = nullptr; size_t d_spdif_words_cap = 0;      // DSPI_OUT_SPDIF on launches the latency layout does not serve: the chain's pair words of one row chunk
    // PDM sub output (dspi_pdm.hip): modulator state per stream, allocated on first use; staging for host buffers
    uint32_t *d_pdm = nullptr;
    int32_t *d_pdm_in = nullptr; size_t d_pdm_in_cap = 0;
    uint32_t *d_pdm_out = nullptr; size_t d_pdm_out_cap = 0;
    int32_t *d_spdif_in = nullptr; size_t d_spdif_in_cap = 0;
    uint32_t *d_spdif_out = nullptr; size_t d_spdif_out_cap = 0;
    std::string err;
};

namespace {

int fail(dspi_ctx *c, int code, const std::string &msg) { if (c) c->err = msg; return code; }

#define HIPCK(c, call)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) return fail((c), DSPI_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

bool valid_stream(const dspi_ctx *c, int32_t s) { return s == DSPI_ALL_STREAMS || (s >= 0 && (uint32_t)s < c->n_streams); }

// parameter object to mutate for `stream` (copy-on-write when shared)
Params *writable(dspi_ctx *c, int32_t stream) {
    int32_t idx = c->stream_image[(size_t)stream];
    if (c->image_refs[(size_t)idx] > 1) {
        // a reference count never drops to zero (the last stream of an image keeps it), so clones always append
        const size_t slot = c->images.size();
        auto clone = std::make_unique<Params>(*c->images[(size_t)idx]);
        clone->dirty = true;
        c->images.push_back(std::move(clone)); c->image_refs.push_back(1);
        c->image_refs[(size_t)idx]--;
        c->stream_image[(size_t)stream] = (int32_t)slot;
        c->assignment_dirty = true;
        idx = (int32_t)slot;
    }
    return c->images[(size_t)idx].get();
}

// A broadcast call may make separate images equal again (the same whole state for everyone, or a request that removes the only
// difference): it asks for the fold-back pass at the next commit.  When that pass finds nothing to fold it leaves the images as they
// were — no re-upload, no rebuilt lists (merge_images) —, so a broadcast volume change on 65 536 distinct presets costs one hashing pass.
template <class F>
int for_targets(dspi_ctx *c, int32_t stream, F f) {
    if (!c) return DSPI_E_INVAL;
    if (!valid_stream(c, stream)) return fail(c, DSPI_E_INVAL, "stream index out of range");
    if (stream == DSPI_ALL_STREAMS) {
        int rc = 0;
        if (c->images.size() > 1) c->merge_hint = true;
        for (size_t i = 0; i < c->images.size(); i++)
            if (c->image_refs[i] > 0) { int r = f(*c->images[i]); if (r != 0) rc = r; }
        return rc;
    }
    return f(*writable(c, stream));
}

// Streams that were given presets of their own and then received the same whole state again (load_bulk / preset / factory reset
// with DSPI_ALL_STREAMS) hold equal parameter objects: fold them back into one image, so the rows return to the shared-parameter
// kernels and a commit is one upload again.  Params is trivially copyable and zero-filled before construction, so equal bytes
// <=> equal parameters, pending state operations included; unequal padding could only keep two images apart, never merge them.
static_assert(std::is_trivially_copyable<Params>::value, "merge_images compares parameter objects as bytes");
void merge_images(dspi_ctx *c) {
    c->merge_hint = false;
    const size_t ni = c->images.size();
    std::vector<size_t> live;
    for (size_t i = 0; i < ni; i++) if (c->image_refs[i] > 0) live.push_back(i);
    if (ni < 2) return;
    std::vector<uint8_t> was_dirty(ni, 0);
    for (size_t i : live) {
        Params &p = *c->images[i];
        was_dirty[i] = p.dirty ? 1 : 0;
        p.dirty = true;                                            // the flag is part of the bytes that are compared; restored below when nothing merges
        if (p.ops.reset_all_eq) memset(p.ops.reset_band, 0, sizeof p.ops.reset_band);      // the wipe covers every band (state_ops_kernel): which single bands a stream's history also marked does not matter
    }
    std::vector<uint64_t> h(ni, 0);
    auto hash = [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; k++) {
            const unsigned char *b = reinterpret_cast<const unsigned char *>(c->images[live[k]].get());
            uint64_t x = 0xcbf29ce484222325ull;
            for (size_t j = 0; j + 8 <= sizeof(Params); j += 8) { uint64_t w; memcpy(&w, b + j, 8); x = (x ^ w) * 0x100000001b3ull; x ^= x >> 29; }
            h[live[k]] = x;
        }
    };
    const size_t n = live.size();
    const size_t nt = n >= 512 ? std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), std::min<size_t>(32, n / 128)) : 1;
    if (nt > 1) {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; t++) th.emplace_back(hash, n * t / nt, n * (t + 1) / nt);
        for (auto &x : th) x.join();
    } else hash(0, n);
    std::unordered_map<uint64_t, std::vector<size_t>> seen;     // hash -> surviving images (old indices)
    std::vector<int32_t> remap(ni, -1);
    std::vector<size_t> keep;
    for (size_t i : live) {
        auto &cands = seen[h[i]];
        int32_t to = -1;
        for (size_t k : cands) if (memcmp(c->images[k].get(), c->images[i].get(), sizeof(Params)) == 0) { to = remap[k]; break; }
        if (to < 0) { to = (int32_t)keep.size(); keep.push_back(i); cands.push_back(i); }
        remap[i] = to;
    }
    if (keep.size() == ni) {                                      // nothing equal, nothing dead: nothing to upload that was not dirty already
        for (size_t i : live) c->images[i]->dirty = was_dirty[i] != 0;
        return;
    }
    std::vector<std::unique_ptr<Params>> images;
    for (size_t i : keep) images.push_back(std::move(c->images[i]));
    c->images = std::move(images);
    c->image_refs.assign(keep.size(), 0);
    for (auto &si : c->stream_image) { si = remap[(size_t)si]; c->image_refs[(size_t)si]++; }
    // per-image caches describe the old numbering: drop them, every image goes up again
    c->image_flags.clear(); c->image_sig.clear(); c->image_bands.clear(); c->image_touched.clear();
    c->assignment_dirty = true; c->launch_dirty = true;
}

const Params &readable(const dspi_ctx *c, int32_t stream) {
    return *c->images[(size_t)c->stream_image[stream == DSPI_ALL_STREAMS ? 0 : (size_t)stream]];
}

template <class P>
int ensure(dspi_ctx *c, P *&ptr, size_t &cap, size_t bytes) {
    if (bytes <= cap) return 0;
    if (ptr) HIPCK(c, hipFree(ptr));
    ptr = nullptr; cap = 0;
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return fail(c, DSPI_E_NOMEM, "hipMalloc failed (" + std::to_string(bytes) + " bytes)");
    ptr = (P *)p; cap = bytes;
    return 0;
}

int rebuild_assignment(dspi_ctx *c) {
    const size_t ni = c->images.size();
    const uint32_t row = (uint32_t)c->sm.row;
    const bool two = c->flavor != 0;                 // float flavour: two streams per lane
    struct Acc { uint32_t wg; uint64_t m0, m1; };
    std::vector<std::vector<Acc>> acc(ni);
    for (uint32_t s = 0; s < c->n_streams; s++) {
        const size_t im = (size_t)c->stream_image[s];
        const uint32_t wg = s / row, col = s % row;
        auto &v = acc[im];
        if (v.empty() || v.back().wg != wg) v.push_back(Acc{wg, 0ull, 0ull});
        if (two) { if (col & 1u) v.back().m1 |= 1ull << (col >> 1); else v.back().m0 |= 1ull << (col >> 1); }
        else v.back().m0 |= 1ull << col;
    }
    size_t total = 0;
    for (int k = 0; k < 4; k++) { c->image_items[k].assign(ni, {}); c->image_item_offset[k].assign(ni, 0); }
    for (size_t i = 0; i < ni; i++)
        for (const Acc &x : acc[i]) {
            c->image_items[0][i].push_back(WgItem{x.wg, 0u, x.m0, x.m1});
            if (two) {
                const uint64_t both = x.m0 & x.m1, only0 = x.m0 & ~x.m1, only1 = x.m1 & ~x.m0;
                if (both) c->image_items[1][i].push_back(WgItem{x.wg, 0u, both, both});
                if (only0) c->image_items[2][i].push_back(WgItem{x.wg, 0u, only0, 0ull});
                if (only1) c->image_items[3][i].push_back(WgItem{x.wg, 0u, only1, 0ull});
            }
        }
    for (int k = 0; k < 4; k++)
        for (size_t i = 0; i < ni; i++) { c->image_item_offset[k][i] = (uint32_t)total; total += c->image_items[k][i].size(); }
    int rc = ensure(c, c->d_items, c->d_items_cap, total * sizeof(WgItem));
    if (rc) return rc;
    HIPCK(c, hipStreamSynchronize(c->hs));      // no state_ops launch may still be reading the list
    {   // one upload (per-stream presets: tens of thousands of one-item lists)
        std::vector<WgItem> all;
        all.reserve(total);
        for (int k = 0; k < 4; k++)
            for (size_t i = 0; i < ni; i++) all.insert(all.end(), c->image_items[k][i].begin(), c->image_items[k][i].end());
        if (!all.empty()) HIPCK(c, hipMemcpy(c->d_items, all.data(), all.size() * sizeof(WgItem), hipMemcpyHostToDevice));
    }
    c->assignment_dirty = false;
    c->launch_dirty = true;
    return 0;
}

dspi_ctx::ImageSig make_sig(const DevImage &img) {
    dspi_ctx::ImageSig g;
    memset(&g, 0, sizeof g);
    g.flags = img.flags; g.ch_bypassed = img.ch_bypassed; g.out_enabled = img.out_enabled; g.out_mute = img.out_mute;
    g.fs_hz = img.fs_hz; g.mute_transition = img.mute_transition; g.i2s_pairs = img.i2s_pairs;
    for (int o = 0; o < kMaxOut; o++) {
        g.delay[o] = img.delay_samples[o];
        if (img.mix[0][o].f != 0.0f) g.mix_nz |= 1u << o;
        if (img.mix[1][o].f != 0.0f) g.mix_nz |= 1u << (kMaxOut + o);
    }
    for (int ch = 0; ch < kMaxCh; ch++) for (int b = 0; b < kBands; b++) g.kinds[ch * kBands + b] = (uint8_t)img.eq[ch][b].kind;
    g.kinds[kMaxCh * kBands] = (uint8_t)img.loud[0].kind; g.kinds[kMaxCh * kBands + 1] = (uint8_t)img.loud[1].kind;
    return g;
}

// The latency layout of the float chain (dspi_chain_skew.inc) serves images with the leveller off.  Class 1: no output runs an EQ
// (disabled, muted, every band flat, or the sub in EQ-worker mode: exactly the cases in which output_item_pk skips the band loops) —
// eight stream pairs per workgroup, the outputs frame-parallel.  Class 2: some output does — two pairs per workgroup, every output a
// systolic row of its own.  Class 3: the leveller is on — the same two pairs and output rows, the groups of the workgroup passing frames
// through rings (dspi_chain_skew_lev.inc).
int skew_class(const dspi_ctx::ImageSig &g) {
    if (g.flags & IF_LEVELLER_ON) return 3;
    for (int o = 0; o < kMaxOut; o++) {
        const bool enabled = (g.out_enabled >> o) & 1u, muted = (g.out_mute >> o) & 1u, flat = (g.ch_bypassed >> (2 + o)) & 1u;
        const bool processed = o != kMaxOut - 1 || (g.flags & IF_SUB_ACTIVE);
        if (processed && enabled && !muted && !flat) return 2;
    }
    return 1;
}
// ... for launches that leave the chip underfilled: class 1 up to one of its eight-pair workgroups per CU (84 KB of LDS each), class 2
// up to two of its two-pair workgroups per CU (79 KB each); beyond that the packed kernel's throughput layout wins
// (tools/probe/probe8.hip, tools/bench_skew.py).  DSPI_F32_LAYOUT=skew|packed forces one (tests, development).
uint32_t skew_pair_limit(int device, int cls) {
    if (const char *e = getenv("DSPI_F32_LAYOUT")) { if (!strcmp(e, "skew")) return 0xffffffffu; if (!strcmp(e, "packed")) return 0u; }
    int cus = 0;
    if (device < 0 || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    return (cls == 1 ? 8u : 4u) * (uint32_t)cus;
}

dspi_ctx::BandHash hash_bands(const DevImage &img) {
    uint64_t a = 0xcbf29ce484222325ull, b = 0x9e3779b97f4a7c15ull;      // FNV-1a and a multiply-xorshift mix over the same words
    auto feed = [&](uint32_t w) { a = (a ^ w) * 0x100000001b3ull; b = (b + w) * 0xff51afd7ed558ccdull; b ^= b >> 29; };
    for (int ch = 0; ch < kMaxCh; ch++) for (int k = 0; k < kBands; k++) for (int j = 0; j < 6; j++) feed(img.eq[ch][k].c[j].u);
    for (int k = 0; k < 2; k++) for (int j = 0; j < 6; j++) feed(img.loud[k].c[j].u);
    return dspi_ctx::BandHash{a, b};
}

int rebuild_launch_lists(dspi_ctx *c) {
    // lists 0 (Q28) and 1 (float lanes whose two streams share an image): one item per (row, image), grouped by leveller
    // on/off for the float kernel variants.  List 2 (float lanes with ONE stream of an image — image_items 2 / 3 = first /
    // second stream, WgItem::image = which; Q28: rows holding several images): the per-lane parameter kernels read every
    // lane's own image, so all images of a row merge into one item.  Launch list 3 stays empty.
    size_t total = 0;
    // float: rows holding several images of ONE structure -> per-lane-value rows (launch list 3, packed kernel + value tile)
    struct RowAcc { uint64_t m0 = 0, m1 = 0; int n = 0; uint32_t first = 0; bool same = true, same_bands = true; };
    std::map<uint32_t, RowAcc> rows_f;
    c->row_pv.assign(c->n_wg, 0);
    if (c->flavor && c->image_sig.size() >= c->images.size()) {
        for (size_t i = 0; i < c->images.size(); i++)
            for (const WgItem &it : c->image_items[0][i]) {
                RowAcc &r = rows_f[it.wg];
                if (r.n++ == 0) r.first = (uint32_t)i;
                else {
                    if (memcmp(&c->image_sig[r.first], &c->image_sig[i], sizeof(dspi_ctx::ImageSig)) != 0) r.same = false;
                    if (c->image_bands[r.first].a != c->image_bands[i].a || c->image_bands[r.first].b != c->image_bands[i].b) r.same_bands = false;
                }
                r.m0 |= it.mask; r.m1 |= it.mask1;
            }
        for (const auto &r : rows_f)
            if (r.second.n > 1 && r.second.same && (r.second.m0 & r.second.m1)) c->row_pv[r.first] = r.second.same_bands ? 2 : 1;
        // "identical filters" (row_pv 2: the kernel takes the band coefficients of the row's first image for every stream) was decided
        // on two 64-bit hashes; back it with the words themselves before it can cost bit-exactness
        std::unique_ptr<DevImage> ref(new DevImage), cur(new DevImage);
        uint32_t ref_row = 0xffffffffu;
        for (size_t i = 0; i < c->images.size(); i++)
            for (const WgItem &it : c->image_items[0][i]) {
                if (c->row_pv[it.wg] != 2) continue;
                const RowAcc &r = rows_f[it.wg];
                if (r.first == i) continue;
                if (ref_row != it.wg) { c->images[r.first]->build_image(*ref); ref_row = it.wg; }
                c->images[i]->build_image(*cur);
                if (memcmp(ref->eq, cur->eq, sizeof ref->eq) != 0 || memcmp(ref->loud, cur->loud, sizeof ref->loud) != 0) c->row_pv[it.wg] = 1;
            }
    }
    for (int lev = 0; lev < 2; lev++) for (int k = 5; k <= 8; k++) c->launch_items[lev][k].clear();
    for (int lev = 0; lev < 2; lev++)
        for (int k = 0; k < 5; k++) {
            auto &v = c->launch_items[lev][k];
            v.clear();
            if (k >= 3) {      // list 3: per-lane values incl. band coefficients; list 4: identical filters, the other numbers per lane
                for (const auto &r : rows_f) {
                    if (c->row_pv[r.first] != (k == 3 ? 1 : 2)) continue;
                    const int ilev = (c->image_flags[r.second.first] & IF_LEVELLER_ON) ? 1 : 0;
                    if (ilev == lev) v.push_back(WgItem{r.first, r.second.first, r.second.m0 & r.second.m1, 0ull});
                }
            } else if (k >= 2) {
                // float: lanes with one stream of an image (k = 2 first, 3 second stream); Q28 (k = 2): rows that hold
                // several images.  Per-lane parameter kernel: all images of a row merge into one item.
                // Float: both lane components go into list 2 (WgItem::image = component), one launch.
                if (lev == 0 && k == 2) {
                    for (int comp = 0; comp < (c->flavor ? 2 : 1); comp++) {
                        std::map<uint32_t, std::pair<uint64_t, int>> rows;      // row -> (lane mask, number of images)
                        const int src = c->flavor ? 2 + comp : 0;
                        for (size_t i = 0; i < c->images.size(); i++)
                            if (c->image_refs[i] > 0)
                                for (const WgItem &it : c->image_items[src][i]) {
                                    if (c->flavor && c->row_pv[it.wg]) continue;      // per-lane-value row: only its half-filled lanes come here (below)
                                    auto &r = rows[it.wg]; r.first |= it.mask; r.second++;
                                }
                        if (c->flavor)
                            for (const auto &rf : rows_f) {
                                if (!c->row_pv[rf.first]) continue;
                                const uint64_t only = comp ? (rf.second.m1 & ~rf.second.m0) : (rf.second.m0 & ~rf.second.m1);
                                if (only) { auto &r = rows[rf.first]; r.first |= only; r.second++; }
                            }
                        for (const auto &r : rows)
                            if (c->flavor || r.second.second > 1) v.push_back(WgItem{r.first, (uint32_t)comp, r.second.first, 0ull});
                    }
                }
            } else {
                std::map<uint32_t, int> per_row;       // Q28: rows with one image keep the workgroup-uniform path
                if (!c->flavor && k == 0)
                    for (size_t i = 0; i < c->images.size(); i++)
                        if (c->image_refs[i] > 0)
                            for (const WgItem &it : c->image_items[0][i]) per_row[it.wg]++;
                for (size_t i = 0; i < c->images.size(); i++) {
                    if (c->image_refs[i] == 0) continue;
                    const int ilev = (c->flavor && (c->image_flags[i] & IF_LEVELLER_ON)) ? 1 : 0;
                    if (ilev != lev) continue;
                    for (WgItem it : c->image_items[k][i]) {
                        if (!c->flavor && k == 0 && per_row[it.wg] > 1) continue;
                        if (c->flavor && k == 1 && c->row_pv[it.wg]) continue;
                        it.image = (uint32_t)i; v.push_back(it);
                    }
                }
            }
            // by row: a range of rows is then a contiguous run of every list (dspi_process stages host buffers row chunk by row chunk)
            std::stable_sort(v.begin(), v.end(), [](const WgItem &x, const WgItem &y) { return x.wg < y.wg; });
        }
    // float, shared-preset lanes with the leveller off: those whose image suits the latency layout move to list 5 / 6 (by class) when
    // the launch is small enough to leave the chip underfilled
    if (c->flavor && c->image_sig.size() >= c->images.size()) {
        uint64_t pairs[4] = {0, 0, 0, 0};
        for (int lev = 0; lev < 2; lev++)
            for (const WgItem &it : c->launch_items[lev][1]) pairs[skew_class(c->image_sig[it.image])] += (uint64_t)__builtin_popcountll(it.mask);
        // the context's last stream when the stream count is odd: a lane that holds one stream.  The packed kernel leaves such lanes to the
        // one-stream kernel (list 2); the latency layout serves them (its stores check the second stream) — a context of ONE stream is this case
        const bool odd = (c->n_streams & 1u) != 0;
        const uint32_t last = c->n_streams - 1u, h_row = last / (uint32_t)c->sm.row, h_lane = (last % (uint32_t)c->sm.row) / 2u;
        const uint32_t h_img = odd ? c->stream_image[last] : 0u;
        const int h_cls = (odd && !c->row_pv[h_row]) ? skew_class(c->image_sig[h_img]) : 0;
        if (h_cls) pairs[h_cls]++;
        bool take[4] = {false, false, false, false};
        for (int cls = 1; cls <= 3; cls++) take[cls] = pairs[cls] > 0 && pairs[cls] <= skew_pair_limit(c->device, cls);
        // The whole context small — the lanes of every class within its limit: EVERY float lane takes the latency layout, whatever the
        // presets: no per-lane-value tiles, no one-stream kernel, any mix of structures.  A workgroup = one part of a row (8 or 2 stream
        // pairs).  The images that hold streams there: one -> a shared-preset item; several of ONE structure (ImageSig) -> one paired-preset
        // item (lists 7 / 8: the kernel reads every slot's numbers from its own image, args.stream_image; the item names the first image, for
        // the structure); several structures -> one item per image, the other images' slots inactive (the kernels store per half).  The
        // limit counts lanes, a lane of the last kind once per image.  DSPI_SKEW_PAIRED=0 keeps to the last form (development, tests).
        {
            const char *ppe = getenv("DSPI_SKEW_PAIRED");
            const bool pp_on = !(ppe && !strcmp(ppe, "0"));
            struct Slot { uint32_t image; uint64_t m0, m1; };
            struct Cell { std::vector<Slot> v; bool same = false; };
            std::map<std::pair<uint32_t, uint32_t>, Cell> cells[4];      // [class]: (row, part) -> images
            uint64_t slots[4] = {0, 0, 0, 0};
            {   // a first bound: the lanes in use, whatever their images
                std::map<uint32_t, uint64_t> used[4];
                for (size_t i = 0; i < c->images.size(); i++)
                    if (c->image_refs[i] > 0)
                        for (const WgItem &it : c->image_items[0][i]) used[skew_class(c->image_sig[i])][it.wg] |= it.mask | it.mask1;
                for (int cls = 1; cls <= 3; cls++) for (const auto &u : used[cls]) slots[cls] += (uint64_t)__builtin_popcountll(u.second);
            }
            bool all_small = slots[1] + slots[2] + slots[3] > 0;
            for (int cls = 1; cls <= 3; cls++) if (slots[cls] > (uint64_t)skew_pair_limit(c->device, cls) * (cls == 2 ? 2u : 1u)) all_small = false;      // (class 2: see below)
            if (all_small) {
                for (size_t i = 0; i < c->images.size(); i++) {
                    if (c->image_refs[i] == 0) continue;
                    const int cls = skew_class(c->image_sig[i]);
                    for (const WgItem &it : c->image_items[0][i])
                        for (uint32_t ppw = cls == 1 ? 8u : 2u, part = 0; part < 64u / ppw; part++) {
                            const uint64_t pm = ((1ull << ppw) - 1ull) << (part * ppw);
                            if ((it.mask | it.mask1) & pm) cells[cls][{it.wg, part}].v.push_back(Slot{(uint32_t)i, it.mask & pm, it.mask1 & pm});
                        }
                }
                for (int cls = 1; cls <= 3; cls++) {
                    slots[cls] = 0;
                    size_t paired = 0;
                    for (auto &cell : cells[cls]) {
                        std::vector<Slot> &v = cell.second.v;
                        bool same = pp_on && v.size() > 1;
                        for (size_t j = 1; same && j < v.size(); j++)
                            if (memcmp(&c->image_sig[v[0].image], &c->image_sig[v[j].image], sizeof(dspi_ctx::ImageSig)) != 0) same = false;
                        cell.second.same = same;
                        paired += same ? 1 : 0;
                        uint64_t u = 0;
                        for (const Slot &sl : v) { if (same) u |= sl.m0 | sl.m1; else slots[cls] += (uint64_t)__builtin_popcountll(sl.m0 | sl.m1); }
                        slots[cls] += (uint64_t)__builtin_popcountll(u);
                    }
                    // Presets with output EQ and no leveller, mostly paired workgroups (every stream its own preset): the alternative is the packed
                    // per-lane-filter kernel on an underfilled chip, and the layout wins up to twice its shared-preset limit (4 096 distinct
                    // presets: 13.9 against 22.6 ms per 200 packets, profiles/r04_small_contexts_per_stream_leveller_off.jsonl).
                    if (slots[cls] > (uint64_t)skew_pair_limit(c->device, cls) * ((cls == 2 && paired * 2 > cells[cls].size()) ? 2u : 1u)) all_small = false;
                }
            }
            if (all_small) {
                for (int lev = 0; lev < 2; lev++) for (int k = 1; k <= 4; k++) c->launch_items[lev][k].clear();
                for (int cls = 1; cls <= 3; cls++)
                    for (const auto &cell : cells[cls]) {
                        const uint32_t row = cell.first.first, part = cell.first.second;
                        const std::vector<Slot> &v = cell.second.v;
                        if (cell.second.same) {
                            uint64_t m0 = 0, m1 = 0;
                            for (const Slot &sl : v) { m0 |= sl.m0; m1 |= sl.m1; }
                            c->launch_items[cls == 3 ? 1 : 0][cls == 2 ? 8 : 7].push_back(WgItem{row, v[0].image | (part << 26), m0, m1});
                        } else
                            for (const Slot &sl : v) c->launch_items[cls == 3 ? 1 : 0][cls == 2 ? 6 : 5].push_back(WgItem{row, sl.image | (part << 26), sl.m0, sl.m1});
                    }
                for (int lev = 0; lev < 2; lev++)
                    for (int k = 5; k <= 8; k++)
                        std::stable_sort(c->launch_items[lev][k].begin(), c->launch_items[lev][k].end(), [](const WgItem &x, const WgItem &y) { return x.wg < y.wg; });
                take[1] = take[2] = take[3] = false;      // (nothing left for the shared-preset rule below)
            }
        }
        // a latency-layout item is ONE workgroup: a row's item is cut into its non-empty parts (8 or 2 stream pairs each), the part rides in
        // the image field's top bits (kSkPartShift = 26, dspi_chain_skew.inc); mask / mask1 = the lanes whose first / second stream take part
        auto push_parts = [&](std::vector<WgItem> &dst, uint32_t row, uint32_t image, uint64_t m0, uint64_t m1, int cls) {
            const uint32_t ppw = cls == 1 ? 8u : 2u;
            for (uint32_t part = 0; part < 64u / ppw; part++)
                if (((m0 | m1) >> (part * ppw)) & ((1ull << ppw) - 1ull)) dst.push_back(WgItem{row, image | (part << 26), m0, m1});
        };
        for (int lev = 0; lev < 2; lev++) {
            auto &v = c->launch_items[lev][1];
            std::vector<WgItem> keep;
            for (const WgItem &it : v) {
                const int cls = skew_class(c->image_sig[it.image]);
                // (class 3 sits in the leveller-on lists: list 5 there is the third shape)
                if (take[cls]) push_parts(c->launch_items[lev][cls == 2 ? 6 : 5], it.wg, it.image, it.mask, it.mask, cls);
                else keep.push_back(it);
            }
            v.swap(keep);
        }
        if (h_cls && take[h_cls]) {
            auto &l2 = c->launch_items[0][2];
            for (size_t i = 0; i < l2.size(); i++)
                if (l2[i].wg == h_row && l2[i].image == 0u && ((l2[i].mask >> h_lane) & 1ull)) {      // (list 2: image = lane component, 0 = first stream)
                    l2[i].mask &= ~(1ull << h_lane);
                    if (l2[i].mask == 0) l2.erase(l2.begin() + (long)i);
                    auto &dst = c->launch_items[h_cls == 3 ? 1 : 0][h_cls == 2 ? 6 : 5];
                    const uint32_t ppw = h_cls == 1 ? 8u : 2u, h_part = h_lane / ppw;
                    bool merged = false;
                    for (WgItem &d : dst) if (d.wg == h_row && d.image == (h_img | (h_part << 26))) { d.mask |= 1ull << h_lane; merged = true; break; }
                    if (!merged) {
                        dst.push_back(WgItem{h_row, h_img | (h_part << 26), 1ull << h_lane, 0ull});
                        std::stable_sort(dst.begin(), dst.end(), [](const WgItem &x, const WgItem &y) { return x.wg < y.wg; });
                    }
                    break;
                }
        }
    }
    for (int lev = 0; lev < 2; lev++)
        for (int k = 0; k < 9; k++) {
            c->launch_item_offset[lev][k] = (uint32_t)total;
            total += c->launch_items[lev][k].size();
        }
    int rc = ensure(c, c->d_litems, c->d_litems_cap, total * sizeof(WgItem));
    if (rc) return rc;
    if ((rc = ensure(c, c->d_stream_image, c->d_stream_image_cap, (size_t)c->n_streams * 4))) return rc;
    HIPCK(c, hipStreamSynchronize(c->hs));      // no launch may still be reading the lists we overwrite
    for (int lev = 0; lev < 2; lev++)
        for (int k = 0; k < 9; k++)
            if (!c->launch_items[lev][k].empty())
                HIPCK(c, hipMemcpy(c->d_litems + c->launch_item_offset[lev][k], c->launch_items[lev][k].data(),
                                   c->launch_items[lev][k].size() * sizeof(WgItem), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(c->d_stream_image, c->stream_image.data(), (size_t)c->n_streams * 4, hipMemcpyHostToDevice));
    // every per-lane-value row gets its tile rebuilt (commit_params)
    for (size_t i = 0; i < c->image_touched.size(); i++) c->image_touched[i] = 1;
    c->launch_dirty = false;
    return 0;
}

bool ops_pending(const StateOps &o) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&o);
    for (size_t i = 0; i < sizeof(StateOps) / 4; i++) if (w[i]) return true;
    return false;
}

// upload dirty images and run pending state mutations; everything is ordered on c->hs
int commit_params(dspi_ctx *c) {
    if (c->merge_hint) merge_images(c);
    if (c->assignment_dirty) { int rc = rebuild_assignment(c); if (rc) return rc; }
    const size_t ni = c->images.size();
    bool any_dirty = false;
    for (size_t i = 0; i < ni; i++) if (c->images[i]->dirty) any_dirty = true;
    if (any_dirty || ni * sizeof(DevImage) > c->d_images_cap) HIPCK(c, hipStreamSynchronize(c->hs));   // no launch may still be reading an image we overwrite
    if (ni * sizeof(DevImage) > c->d_images_cap) {
        DevImage *old = c->d_images; size_t oldcap = c->d_images_cap;
        c->d_images = nullptr; c->d_images_cap = 0;
        int rc = ensure(c, c->d_images, c->d_images_cap, (ni + 8) * sizeof(DevImage));
        if (rc) return rc;
        if (old) { HIPCK(c, hipMemcpy(c->d_images, old, oldcap, hipMemcpyDeviceToDevice)); HIPCK(c, hipFree(old)); }
        for (auto &p : c->images) p->dirty = true;
    }
    // dirty images go up in contiguous runs (per-stream presets dirty thousands at once)
    if (c->image_flags.size() < ni) c->image_flags.resize(ni, 0u);
    if (c->flavor && c->image_sig.size() < ni) {
        dspi_ctx::ImageSig none; memset(&none, 0xff, sizeof none);
        c->image_sig.resize(ni, none); c->image_bands.resize(ni, dspi_ctx::BandHash{0, 0}); c->image_touched.resize(ni, 1); c->launch_dirty = true;
    }
    // a run of dirty images: built on all host threads when it is long (every stream its own preset: tens of thousands at once),
    // compared with what the launch lists were built from, uploaded with one copy
    std::vector<DevImage> run;
    std::vector<dspi_ctx::ImageSig> run_sig;
    std::vector<dspi_ctx::BandHash> run_bands;
    for (size_t i = 0; i < ni;) {
        if (!c->images[i]->dirty) { i++; continue; }
        size_t j = i;
        while (j < ni && c->images[j]->dirty) j++;
        const size_t n = j - i;
        run.resize(n);
        if (c->flavor) { run_sig.resize(n); run_bands.resize(n); }
        const unsigned csr = _mm_getcsr();      // the workers round and flush exactly like the calling thread
        auto build = [&](size_t k0, size_t k1) {
            _mm_setcsr(csr);
            for (size_t k = k0; k < k1; k++) {
                c->images[i + k]->build_image(run[k]);
                if (c->flavor) { run_sig[k] = make_sig(run[k]); run_bands[k] = hash_bands(run[k]); }
            }
        };
        const size_t nt = n >= 512 ? std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), std::min<size_t>(32, n / 128)) : 1;
        if (nt > 1) {
            std::vector<std::thread> th;
            for (size_t t = 0; t < nt; t++) th.emplace_back(build, n * t / nt, n * (t + 1) / nt);
            for (auto &x : th) x.join();
        } else build(0, n);
        for (size_t k = 0; k < n; k++) {
            const size_t ii = i + k;
            if ((c->image_flags[ii] ^ run[k].flags) & IF_LEVELLER_ON) c->launch_dirty = true;
            c->image_flags[ii] = run[k].flags;
            for (const float al : {run[k].lv_alpha_attack, run[k].lv_alpha_release}) {
                uint32_t bits; memcpy(&bits, &al, 4);
                if (std::find(c->lv_alphas.begin(), c->lv_alphas.end(), bits) == c->lv_alphas.end()) c->lv_alphas.push_back(bits);
            }
            if (c->flavor) {
                if (memcmp(&run_sig[k], &c->image_sig[ii], sizeof(dspi_ctx::ImageSig)) != 0) { c->image_sig[ii] = run_sig[k]; c->launch_dirty = true; }
                if (run_bands[k].a != c->image_bands[ii].a || run_bands[k].b != c->image_bands[ii].b) { c->image_bands[ii] = run_bands[k]; c->launch_dirty = true; }
                c->image_touched[ii] = 1;
            }
            c->images[ii]->dirty = false;
        }
        HIPCK(c, hipMemcpy(c->d_images + i, run.data(), n * sizeof(DevImage), hipMemcpyHostToDevice));
        i = j;
    }
    // pending state mutations: consecutive images with the same mutation share one launch (their workgroup items are
    // consecutive in d_items, list 0)
    for (size_t i = 0; i < ni;) {
        Params &p = *c->images[i];
        if (!ops_pending(p.ops)) { i++; continue; }
        size_t j = i + 1;
        while (j < ni && memcmp(&c->images[j]->ops, &p.ops, sizeof(StateOps)) == 0) j++;
        const uint32_t first = c->image_item_offset[0][i];
        const uint32_t count = (uint32_t)((j < ni ? c->image_item_offset[0][j] : c->image_item_offset[0][ni - 1] + (uint32_t)c->image_items[0][ni - 1].size()) - first);
        if (count)
            HIPCK(c, launch_state_ops(c->flavor, c->d_items + first, count, p.ops, c->d_state, c->d_dlines, c->d_ring, c->n_streams, c->hs));
        for (size_t k = i; k < j; k++) c->images[k]->ops = StateOps{};
        i = j;
    }
    if (c->launch_dirty) { int rc = rebuild_launch_lists(c); if (rc) return rc; }
    if (c->flavor) {       // value tiles of the per-lane-value rows that hold an image uploaded above
        std::vector<uint32_t> rows;
        std::vector<uint8_t> seen(c->n_wg, 0);
        for (size_t i = 0; i < ni; i++) {
            if (!c->image_touched[i]) continue;
            c->image_touched[i] = 0;
            for (const WgItem &it : c->image_items[0][i])
                if (c->row_pv[it.wg] && !seen[it.wg]) { seen[it.wg] = 1; rows.push_back(it.wg); }
        }
        if (!rows.empty()) {
            int rc = ensure(c, c->d_vals, c->d_vals_cap, (size_t)c->n_wg * kPvTileFloats * sizeof(float));
            if (rc) return rc;
            if ((rc = ensure(c, c->d_pv_rows, c->d_pv_rows_cap, (size_t)c->n_wg * 4))) return rc;
            HIPCK(c, hipStreamSynchronize(c->hs));      // no launch may still be reading the tiles or the row list
            HIPCK(c, hipMemcpy(c->d_pv_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
            static const bool all_differ = getenv("DSPI_DEBUG") && (strtoul(getenv("DSPI_DEBUG"), nullptr, 0) & 1u);      // development switch (timing of the worst case)
            HIPCK(c, launch_pv_build(c->d_images, c->d_stream_image, c->d_pv_rows, (uint32_t)rows.size(), c->d_vals, c->n_streams, all_differ, c->hs));
        }
    }
    return 0;
}

int read_stream_words(dspi_ctx *c, uint32_t stream, int slot0, int count, uint32_t *out) {
    const uint32_t row = (uint32_t)c->sm.row, wg = stream / row, col = stream % row;
    const uint32_t *src = c->d_state + ((size_t)wg * c->sm.n_slots + slot0) * row + col;
    HIPCK(c, hipSetDevice(c->device));            // several contexts on different GPUs may share the process
    HIPCK(c, hipStreamSynchronize(c->hs));
    HIPCK(c, hipMemcpy2D(out, 4, src, (size_t)row * 4, 4, (size_t)count, hipMemcpyDeviceToHost));
    return 0;
}

int fetch_status(dspi_ctx *c, int32_t stream, uint16_t *peaks, uint16_t *clip) {
    for (int i = 0; i < kMaxCh; i++) peaks[i] = 0;
    *clip = 0;
    if (c->device == DSPI_DEVICE_NONE) return 0;
    const uint32_t s = stream == DSPI_ALL_STREAMS ? 0u : (uint32_t)stream;
    uint32_t w[kMaxCh + 4];
    int rc = read_stream_words(c, s, c->sm.peaks, c->sm.n_ch + 4, w);
    if (rc) return rc;
    for (int i = 0; i < c->sm.n_ch; i++) peaks[i] = (uint16_t)w[i];
    *clip = (uint16_t)(w[c->sm.n_ch] | w[c->sm.n_ch + 1] | w[c->sm.n_ch + 2] | w[c->sm.n_ch + 3]);
    return 0;
}

int zero_clips(dspi_ctx *c, int32_t stream) {
    if (c->device == DSPI_DEVICE_NONE) return 0;
    HIPCK(c, hipSetDevice(c->device));
    HIPCK(c, hipStreamSynchronize(c->hs));
    const size_t row = (size_t)c->sm.row, pitch = (size_t)c->sm.n_slots * row * 4;
    if (stream == DSPI_ALL_STREAMS) {
        HIPCK(c, hipMemset2D(c->d_state + (size_t)c->sm.clip * row, pitch, 0, 4 * row * 4, c->n_wg));
    } else {
        const uint32_t wg = (uint32_t)stream / (uint32_t)row, col = (uint32_t)stream % (uint32_t)row;
        uint32_t z[4] = {0, 0, 0, 0};
        HIPCK(c, hipMemcpy2D(c->d_state + ((size_t)wg * c->sm.n_slots + c->sm.clip) * row + col, row * 4, z, 4, 4, 4, hipMemcpyHostToDevice));
    }
    return 0;
}

}  // namespace

extern "C" {

int dspi_abi_version(void) { return DSPI_ABI_VERSION; }

int dspi_create(dspi_ctx **out, int flavor, uint32_t n_streams, int hip_device) {
    const bool fma = (flavor & DSPI_FLOAT_CONTRACT_FMA) != 0, populated = (flavor & DSPI_BOOT_POPULATED_FLASH) != 0;
    flavor &= ~(DSPI_FLOAT_CONTRACT_FMA | DSPI_BOOT_POPULATED_FLASH);
    if (!out || (flavor != DSPI_FLAVOR_RP2040_Q28 && flavor != DSPI_FLAVOR_RP2350_F32) || n_streams == 0) return DSPI_E_INVAL;
    if (fma && flavor != DSPI_FLAVOR_RP2350_F32) return DSPI_E_INVAL;      // the RP2040 has no FPU: nothing to contract
    dspi_ctx *c = new (std::nothrow) dspi_ctx();
    if (!c) return DSPI_E_NOMEM;
    c->flavor = flavor;
    c->n_streams = n_streams;
    c->device = hip_device;
    c->sm = make_state_map(flavor);
    c->n_wg = (n_streams + (uint32_t)c->sm.row - 1) / (uint32_t)c->sm.row;
    c->fma = fma;
    c->populated = populated;
    c->no_direct = getenv("DSPI_NO_DIRECT") != nullptr;
    if (const char *e = getenv("DSPI_DIRECT_SPIN_US")) { const long v = atol(e); if (v > 0) c->direct_spin_us = (uint32_t)std::min<long>(v, 1000000L); }
    c->images.push_back(std::make_unique<Params>(flavor, fma, !populated));
    c->image_refs.push_back(n_streams);
    c->stream_image.assign(n_streams, 0);
    *out = c;
    if (hip_device == DSPI_DEVICE_NONE) return DSPI_OK;

    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { *out = nullptr; delete c; return DSPI_E_NODEVICE; }
    auto bail = [&](int code) { dspi_destroy(c); *out = nullptr; return code; };
    if (hipSetDevice(hip_device) != hipSuccess) return bail(DSPI_E_NODEVICE);
    if (hipStreamCreateWithFlags(&c->hs, hipStreamNonBlocking) != hipSuccess) return bail(DSPI_E_HIP);
    const size_t row = (size_t)c->sm.row;
    const size_t state_b = (size_t)c->n_wg * c->sm.n_slots * row * 4;
    const size_t dl_b = (size_t)c->n_wg * c->sm.n_out * (size_t)c->sm.max_delay * row * 4;
    const size_t ring_b = (size_t)c->n_wg * kRingLen * 2 * row * 4;
    if (hipMalloc((void **)&c->d_state, state_b) != hipSuccess || hipMalloc((void **)&c->d_dlines, dl_b) != hipSuccess ||
        hipMalloc((void **)&c->d_ring, ring_b) != hipSuccess)
        return bail(DSPI_E_NOMEM);
    if (hipMemsetAsync(c->d_state, 0, state_b, c->hs) != hipSuccess || hipMemsetAsync(c->d_dlines, 0, dl_b, c->hs) != hipSuccess ||
        hipMemsetAsync(c->d_ring, 0, ring_b, c->hs) != hipSuccess || launch_state_init(flavor, c->d_state, c->n_wg, c->hs) != hipSuccess ||
        hipStreamSynchronize(c->hs) != hipSuccess)
        return bail(DSPI_E_HIP);
    return DSPI_OK;
}

void dspi_destroy(dspi_ctx *c) {
    if (!c) return;
    if (c->device != DSPI_DEVICE_NONE) {
        (void)hipSetDevice(c->device);
        if (c->hs) (void)hipStreamSynchronize(c->hs);
        for (void *p : {(void *)c->d_state, (void *)c->d_dlines, (void *)c->d_ring, (void *)c->d_xwords, (void *)c->d_vals, (void *)c->d_pv_rows, (void *)c->d_images, (void *)c->d_items, (void *)c->d_litems, (void *)c->d_stream_image, (void *)c->d_pdm, (void *)c->d_pdm_in, (void *)c->d_pdm_out, (void *)c->d_spdif_in, (void *)c->d_spdif_out, c->d_in,
                        (void *)c->d_pairs, (void *)c->d_sub, (void *)c->d_peaks, (void *)c->d_clip, (void *)c->d_spdif_words})
            if (p) (void)hipFree(p);
        if (c->h_direct) (void)hipHostFree(c->h_direct);
        if (c->h_done) (void)hipHostFree(c->h_done);
        for (hipEvent_t e : c->pipe_events) (void)hipEventDestroy(e);
        if (c->hs_in) (void)hipStreamDestroy(c->hs_in);
        if (c->hs_out) (void)hipStreamDestroy(c->hs_out);
        if (c->hs) (void)hipStreamDestroy(c->hs);
    }
    delete c;
}

const char *dspi_last_error(const dspi_ctx *c) { return c ? c->err.c_str() : "null context"; }
int dspi_num_channels(const dspi_ctx *c) { return c ? c->sm.n_ch : DSPI_E_INVAL; }
int dspi_num_outputs(const dspi_ctx *c) { return c ? c->sm.n_out : DSPI_E_INVAL; }
int dspi_num_pairs(const dspi_ctx *c) { return c ? c->sm.n_pairs : DSPI_E_INVAL; }
uint32_t dspi_num_streams(const dspi_ctx *c) { return c ? c->n_streams : 0; }
uint32_t dspi_tile_streams(const dspi_ctx *c) { return c ? (uint32_t)c->sm.row : 0; }
void *dspi_hip_stream(dspi_ctx *c) { return c ? (void *)c->hs : nullptr; }

int dspi_factory_defaults(dspi_ctx *c, int32_t stream) {
    return for_targets(c, stream, [](Params &p) { p.factory_reset(); return 0; });
}
int dspi_load_bulk(dspi_ctx *c, int32_t stream, const void *blob, size_t len) {
    if (!blob) return DSPI_E_INVAL;
    return for_targets(c, stream, [&](Params &p) { return p.load_bulk(blob, len); });
}
int dspi_collect_bulk(dspi_ctx *c, int32_t stream, void *blob, size_t cap) {
    if (!c || !blob || !valid_stream(c, stream)) return DSPI_E_INVAL;
    return readable(c, stream).collect_bulk(blob, cap);
}
int dspi_load_preset_slot(dspi_ctx *c, int32_t stream, const void *image, size_t len, int expect_slot) {
    if (!image) return DSPI_E_INVAL;
    return for_targets(c, stream, [&](Params &p) { return p.load_slot(image, len, expect_slot); });
}
int dspi_load_flash_dump(dspi_ctx *c, int32_t stream, const void *dump, size_t len) {
    if (!dump) return DSPI_E_INVAL;
    if (len < kFlashDumpBytes) return DSPI_E_SHORT;
    // A context of devices with a populated flash (DSPI_BOOT_POPULATED_FLASH) that has not processed audio yet BOOTS from the dump
    // (preset_boot_load -> apply_slot_to_live, flash_storage.c:1047-1082: no mute, no line zeroing): exact against the firmware from
    // frame 0.  Every other context is a running device that switches to the preset the dump selects (preset_load, :794-849).
    const bool as_boot = c->populated && !c->audio_started;
    return for_targets(c, stream, [&](Params &p) { return p.load_flash_dump(dump, len, as_boot); });
}
int dspi_flash_read_directory(const void *dump, size_t len, dspi_flash_dir *out) {
    if (!dump || !out) return DSPI_E_INVAL;
    FlashDirectory d;
    parse_flash_directory(dump, len, d);
    memset(out, 0, sizeof *out);
    out->valid = d.valid; out->version = d.version;
    out->startup_mode = d.startup_mode; out->default_slot = d.default_slot; out->last_active_slot = d.last_active_slot; out->include_pins = d.include_pins;
    out->slot_occupied = d.slot_occupied; out->master_volume_mode = d.master_volume_mode; out->master_volume_db = d.master_volume_db;
    memcpy(out->slot_names, d.slot_names, sizeof out->slot_names);
    return DSPI_OK;
}
int dspi_save_preset_slot(dspi_ctx *c, int32_t stream, void *image, size_t cap, int slot_index) {
    if (!c || !image || !valid_stream(c, stream)) return DSPI_E_INVAL;
    return readable(c, stream).save_slot(image, cap, slot_index);
}
int dspi_vendor_set(dspi_ctx *c, int32_t stream, uint8_t req, uint16_t wValue, const void *payload, uint16_t len) {
    if (len && !payload) return DSPI_E_INVAL;
    return for_targets(c, stream, [&](Params &p) { return p.vendor_set(req, wValue, payload, len); });
}
int dspi_vendor_get(dspi_ctx *c, int32_t stream, uint8_t req, uint16_t wValue, void *buf, uint16_t cap) {
    if (!c || !buf || !valid_stream(c, stream)) return DSPI_E_INVAL;
    uint16_t peaks[kMaxCh], clip = 0;
    const bool needs_status = (req == 0x50 || req == 0x83);
    if (needs_status) { int rc = fetch_status(c, stream, peaks, &clip); if (rc) return rc; }
    const uint16_t before = clip;
    int n;
    if (req == 0x53) {   // REQ_FACTORY_RESET answers with a status byte and mutates state
        int rc = dspi_factory_defaults(c, stream);
        if (rc) return rc;
        if (cap < 1) return DSPI_E_SHORT;
        *(uint8_t *)buf = 0;
        return 1;
    }
    if (req == 0xC0) {   // REQ_SET_OUTPUT_TYPE: answered from a copy unless it really changes a slot's type (no clone, no mute for a no-op or a refusal)
        if (cap < 1) return DSPI_E_SHORT;
        const uint8_t slot = wValue & 0xFF, type = (wValue >> 8) & 0xFF;
        bool changes = false;
        for (size_t i = 0; i < c->images.size() && !changes; i++) {
            if (stream == DSPI_ALL_STREAMS ? c->image_refs[i] == 0 : (size_t)c->stream_image[(size_t)stream] != i) continue;
            const Params &r = *c->images[i];
            changes = slot < r.n_pairs && type <= 1 && type != r.output_types[slot];
        }
        if (!changes) { Params view = readable(c, stream); return view.vendor_get(req, wValue, buf, cap, peaks, &clip); }
    }
    if (req == 0xD6 || req == 0xC0) {   // REQ_SAVE_MASTER_VOLUME mutates the directory copy, REQ_SET_OUTPUT_TYPE the slot type (+ pipeline mute)
        return for_targets(c, stream, [&](Params &p) { int r = p.vendor_get(req, wValue, buf, cap, peaks, &clip); return r < 0 ? r : 0; }) == 0 ? 1 : DSPI_E_SHORT;
    }
    Params tmp_view = readable(c, stream);    // GETs never change parameters; work on a copy
    n = tmp_view.vendor_get(req, wValue, buf, cap, peaks, &clip);
    // REQ_CLEAR_CLIPS: with DSPI_ALL_STREAMS the answer carries stream 0's flags (one device answers one request), but every
    // stream's sticky bits are cleared whatever stream 0 held
    if (req == 0x83 && n >= 0 && (before != 0 || stream == DSPI_ALL_STREAMS)) { int rc = zero_clips(c, stream); if (rc) return rc; }
    return n;
}
int dspi_set_host_volume(dspi_ctx *c, int32_t stream, int16_t v) {
    return for_targets(c, stream, [&](Params &p) { p.set_volume(v); return 0; });
}
int dspi_set_mute(dspi_ctx *c, int32_t stream, int mute) {
    return for_targets(c, stream, [&](Params &p) { p.set_mute(mute != 0); return 0; });
}
int dspi_set_sample_rate(dspi_ctx *c, int32_t stream, uint32_t hz) {
    if (hz != 44100 && hz != 48000 && hz != 96000) return DSPI_E_INVAL;
    return for_targets(c, stream, [&](Params &p) { return p.set_rate(hz); });
}

int dspi_get_status(dspi_ctx *c, int32_t stream, void *buf, size_t cap) {
    if (!c || !buf || !valid_stream(c, stream)) return DSPI_E_INVAL;
    const int n = c->sm.n_ch * 2 + 4;
    if (cap < (size_t)n) return DSPI_E_SHORT;
    return dspi_vendor_get(c, stream, 0x50, 9, buf, (uint16_t)n);
}
int dspi_clear_clips(dspi_ctx *c, int32_t stream) {
    if (!c || !valid_stream(c, stream)) return DSPI_E_INVAL;
    uint16_t f = 0;
    int n = dspi_vendor_get(c, stream, 0x83, 0, &f, 2);
    return n < 0 ? n : (int)f;
}

int dspi_debug_image(dspi_ctx *c, int32_t stream, void *buf, size_t cap) {
    if (!c || !buf || !valid_stream(c, stream)) return DSPI_E_INVAL;
    if (cap < sizeof(DevImage)) return DSPI_E_SHORT;
    DevImage img;
    readable(c, stream).build_image(img);
    memcpy(buf, &img, sizeof(img));
    return (int)sizeof(img);
}

int dspi_debug_launch_plan(dspi_ctx *c, uint32_t *counts, size_t n_counts) {
    if (!c || !counts || n_counts < 5) return DSPI_E_INVAL;
    const int n = n_counts >= 7 ? 7 : n_counts >= 6 ? 6 : 5;      // [5]: items of the latency layout (dspi_chain_skew.inc), [6]: those of them with paired presets
    for (int k = 0; k < 5; k++) counts[k] = (uint32_t)(c->launch_items[0][k].size() + c->launch_items[1][k].size());
    if (n >= 6) { counts[5] = 0; for (int k = 5; k <= 8; k++) counts[5] += (uint32_t)(c->launch_items[0][k].size() + c->launch_items[1][k].size()); }      // every shape of the latency layout
    if (n >= 7) { counts[6] = 0; for (int k = 7; k <= 8; k++) counts[6] += (uint32_t)(c->launch_items[0][k].size() + c->launch_items[1][k].size()); }
    if (c->flavor) counts[0] = 0;      // (list 0 of a float context is bookkeeping for the state mutations, never launched)
    return n;
}

int dspi_debug_direct_stats(dspi_ctx *c, uint64_t *out, size_t n) {
    if (!c || !out || n < 5) return DSPI_E_INVAL;
    for (int i = 0; i < 5; i++) out[i] = c->direct_stats[i];
    return 5;
}

int dspi_debug_image_count(dspi_ctx *c) {
    if (!c) return DSPI_E_INVAL;
    if (c->merge_hint) merge_images(c);
    return (int)c->images.size();
}

int dspi_debug_eq_taps(dspi_ctx *c, int32_t stream, int channel, const float *x, uint32_t n, float *taps, float *other) {
    if (!c || !x || !taps || !other || n == 0 || n > (1u << 20) || !valid_stream(c, stream) || stream < 0) return DSPI_E_INVAL;
    if (c->flavor != DSPI_FLAVOR_RP2350_F32 || channel < 0 || channel >= c->sm.n_ch) return DSPI_E_INVAL;
    if (c->device == DSPI_DEVICE_NONE) return fail(c, DSPI_E_NODEVICE, "host-only context: the HIP path is the only audio path");
    HIPCK(c, hipSetDevice(c->device));
    DevImage img;
    readable(c, stream).build_image(img);
    DevImage *d_img = nullptr; float *d_x = nullptr, *d_t = nullptr, *d_o = nullptr;
    const size_t nb = (size_t)n * sizeof(float);
    auto done = [&](int rc) { (void)hipFree(d_img); (void)hipFree(d_x); (void)hipFree(d_t); (void)hipFree(d_o); return rc; };
    if (hipMalloc((void **)&d_img, sizeof img) != hipSuccess || hipMalloc((void **)&d_x, nb) != hipSuccess ||
        hipMalloc((void **)&d_t, nb * (kBands + 1)) != hipSuccess || hipMalloc((void **)&d_o, nb * kBands) != hipSuccess)
        return done(fail(c, DSPI_E_NOMEM, "hipMalloc failed (EQ taps)"));
    if (hipMemcpy(d_img, &img, sizeof img, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_x, x, nb, hipMemcpyHostToDevice) != hipSuccess ||
        launch_eq_taps(c->fma, d_img, channel, d_x, n, d_t, d_o, c->hs) != hipSuccess || hipStreamSynchronize(c->hs) != hipSuccess ||
        hipMemcpy(taps, d_t, nb * (kBands + 1), hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(other, d_o, nb * kBands, hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(c, DSPI_E_HIP, "EQ taps failed"));
    return done(DSPI_OK);
}

int dspi_debug_detmath(dspi_ctx *c, int which, const float *a, const float *b, uint32_t n, float *out) {
    if (!c || !a || !out || ((which == 1 || which == 4) && !b) || n == 0 || n > (1u << 24) || which < 0 || which > 4) return DSPI_E_INVAL;
    if (c->device == DSPI_DEVICE_NONE) return fail(c, DSPI_E_NODEVICE, "host-only context: the HIP path is the only audio path");
    HIPCK(c, hipSetDevice(c->device));
    float *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
    const size_t nb = (size_t)n * sizeof(float);
    auto done = [&](int rc) { (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o); return rc; };
    if (hipMalloc((void **)&d_a, nb) != hipSuccess || hipMalloc((void **)&d_b, nb) != hipSuccess || hipMalloc((void **)&d_o, nb) != hipSuccess)
        return done(fail(c, DSPI_E_NOMEM, "hipMalloc failed (detmath)"));
    if (hipMemcpy(d_a, a, nb, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_b, (which == 1 || which == 4) ? b : a, nb, hipMemcpyHostToDevice) != hipSuccess ||
        launch_detmath(which, d_a, d_b, n, d_o, c->hs) != hipSuccess || hipStreamSynchronize(c->hs) != hipSuccess ||
        hipMemcpy(out, d_o, nb, hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(c, DSPI_E_HIP, "detmath kernel failed"));
    return done(DSPI_OK);
}

// ---- PDM sub output: pdm_generator.c:351-397 per sample (dspi_pdm.hip) ----
static int pdm_state(dspi_ctx *c) {
    if (c->d_pdm) return 0;
    const size_t b = (size_t)c->n_wg * kPdmStateWords * c->sm.row * 4;
    if (hipMalloc((void **)&c->d_pdm, b) != hipSuccess) return fail(c, DSPI_E_NOMEM, "hipMalloc failed (PDM state)");
    HIPCK(c, hipMemsetAsync(c->d_pdm, 0, b, c->hs));
    HIPCK(c, launch_pdm_reset(c->d_pdm, c->n_streams, (uint32_t)c->sm.row, c->n_wg, -1, 1, c->hs));
    return 0;
}

int dspi_pdm_modulate(dspi_ctx *c, const int32_t *sub, uint32_t n_frames, uint32_t *words, uint32_t flags) {
    if (!c || !sub || !words || n_frames == 0) return DSPI_E_INVAL;
    if (c->device == DSPI_DEVICE_NONE) return fail(c, DSPI_E_NODEVICE, "host-only context: the HIP path is the only audio path");
    HIPCK(c, hipSetDevice(c->device));
    int rc = pdm_state(c);
    if (rc) return rc;
    const bool tiled = flags & DSPI_OUT_TILED, dev = flags & DSPI_MEM_DEVICE;
    const size_t cols = tiled ? (size_t)c->n_wg * c->sm.row : (size_t)c->n_streams;
    const size_t in_b = cols * n_frames * 4, out_b = in_b * 8;
    const int32_t *d_in = sub;
    uint32_t *d_out = words;
    if (!dev) {
        if ((rc = ensure(c, c->d_pdm_in, c->d_pdm_in_cap, in_b)) || (rc = ensure(c, c->d_pdm_out, c->d_pdm_out_cap, out_b))) return rc;
        HIPCK(c, hipMemcpyAsync(c->d_pdm_in, sub, in_b, hipMemcpyHostToDevice, c->hs));
        d_in = c->d_pdm_in; d_out = c->d_pdm_out;
    }
    HIPCK(c, launch_pdm(tiled, c->d_pdm, d_in, d_out, c->n_streams, n_frames, (uint32_t)c->sm.row, c->n_wg, c->hs));
    if (!dev) {
        HIPCK(c, hipMemcpyAsync(words, c->d_pdm_out, out_b, hipMemcpyDeviceToHost, c->hs));
        HIPCK(c, hipStreamSynchronize(c->hs));
    }
    return DSPI_OK;
}

int dspi_pdm_restart(dspi_ctx *c, int32_t stream) {
    if (!c || !valid_stream(c, stream)) return DSPI_E_INVAL;
    if (c->device == DSPI_DEVICE_NONE) return DSPI_E_NODEVICE;
    HIPCK(c, hipSetDevice(c->device));
    int rc = pdm_state(c);
    if (rc) return rc;
    HIPCK(c, launch_pdm_reset(c->d_pdm, c->n_streams, (uint32_t)c->sm.row, c->n_wg, stream == DSPI_ALL_STREAMS ? -1 : stream, 0, c->hs));
    return DSPI_OK;
}

// ---- S/PDIF subframes: pico_audio_spdif_multi sample_encoding.h:27-47 + audio_spdif.c:76-116 (dspi_spdif.hip) ----
int dspi_spdif_encode(dspi_ctx *c, const int32_t *pairs, uint32_t n_frames, uint32_t block_pos, uint32_t *subframes, uint32_t flags) {
    if (!c || !pairs || !subframes || n_frames == 0 || block_pos >= 192) return DSPI_E_INVAL;
    if (c->device == DSPI_DEVICE_NONE) return fail(c, DSPI_E_NODEVICE, "host-only context: the HIP path is the only audio path");
    HIPCK(c, hipSetDevice(c->device));
    const bool tiled = flags & DSPI_OUT_TILED, dev = flags & DSPI_MEM_DEVICE;
    const size_t cols = tiled ? (size_t)c->n_wg * c->sm.row : (size_t)c->n_streams;
    const size_t in_b = cols * c->sm.n_pairs * n_frames * 8, out_b = in_b * 2;
    const int32_t *d_in = pairs;
    uint32_t *d_out = subframes;
    int rc;
    if (!dev) {
        if ((rc = ensure(c, c->d_spdif_in, c->d_spdif_in_cap, in_b)) || (rc = ensure(c, c->d_spdif_out, c->d_spdif_out_cap, out_b))) return rc;
        HIPCK(c, hipMemcpyAsync(c->d_spdif_in, pairs, in_b, hipMemcpyHostToDevice, c->hs));
        d_in = c->d_spdif_in; d_out = c->d_spdif_out;
    }
    // the sample-rate byte of the channel status is each device's own (audio_spdif.c:250-256): one word for everybody while every live
    // image runs at one rate, else per stream from the committed images
    const uint32_t fs = readable(c, DSPI_ALL_STREAMS).freq;
    bool one_rate = true;
    for (size_t i = 0; i < c->images.size(); i++) if (c->image_refs[i] > 0 && c->images[i]->freq != fs) one_rate = false;
    SpdifRates rates{nullptr, nullptr, 0u};
    if (!one_rate) {
        if ((rc = commit_params(c))) return rc;
        rates = SpdifRates{c->d_images, c->d_stream_image, 0u};
    }
    HIPCK(c, launch_spdif(tiled, d_in, d_out, c->n_streams, (uint32_t)c->sm.n_pairs, n_frames, (uint32_t)c->sm.row, c->n_wg, block_pos, fs, rates, c->hs));
    if (!dev) {
        HIPCK(c, hipMemcpyAsync(subframes, c->d_spdif_out, out_b, hipMemcpyDeviceToHost, c->hs));
        HIPCK(c, hipStreamSynchronize(c->hs));
    }
    return (int)((block_pos + n_frames) % 192u);
}

// ---- I2S slots: pico_audio_i2s_multi/audio_i2s_multi.c:217-226 (dspi_spdif.hip) ----
int dspi_i2s_encode(dspi_ctx *c, const int32_t *pairs, uint32_t n_frames, uint32_t pair_mask, uint32_t *words, uint32_t flags) {
    if (!c || !pairs || !words || n_frames == 0 || (pair_mask >> c->sm.n_pairs) != 0) return DSPI_E_INVAL;
    if (c->device == DSPI_DEVICE_NONE) return fail(c, DSPI_E_NODEVICE, "host-only context: the HIP path is the only audio path");
    HIPCK(c, hipSetDevice(c->device));
    if (pair_mask == DSPI_I2S_PAIRS_BY_TYPE) {          // the slots whose output type is I2S (output_types[], REQ_SET_OUTPUT_TYPE)
        const Params &p = readable(c, DSPI_ALL_STREAMS);
        for (int i = 0; i < c->sm.n_pairs; i++) if (p.output_types[i] == 1) pair_mask |= 1u << i;
        if (pair_mask == 0) return 0;
    }
    const bool tiled = flags & DSPI_OUT_TILED, dev = flags & DSPI_MEM_DEVICE;
    const size_t cols = tiled ? (size_t)c->n_wg * c->sm.row : (size_t)c->n_streams;
    const size_t bytes = cols * c->sm.n_pairs * n_frames * 8;
    const int32_t *d_in = pairs;
    uint32_t *d_out = words;
    int rc;
    if (!dev) {
        if ((rc = ensure(c, c->d_spdif_in, c->d_spdif_in_cap, bytes)) || (rc = ensure(c, c->d_spdif_out, c->d_spdif_out_cap, bytes))) return rc;
        HIPCK(c, hipMemcpyAsync(c->d_spdif_in, pairs, bytes, hipMemcpyHostToDevice, c->hs));
        HIPCK(c, hipMemcpyAsync(c->d_spdif_out, words, bytes, hipMemcpyHostToDevice, c->hs));     // pairs outside the mask keep the caller's words
        d_in = c->d_spdif_in; d_out = c->d_spdif_out;
    }
    HIPCK(c, launch_i2s(tiled, d_in, d_out, c->n_streams, (uint32_t)c->sm.n_pairs, n_frames, (uint32_t)c->sm.row, c->n_wg, pair_mask, c->hs));
    if (!dev) {
        HIPCK(c, hipMemcpyAsync(words, c->d_spdif_out, bytes, hipMemcpyDeviceToHost, c->hs));
        HIPCK(c, hipStreamSynchronize(c->hs));
    }
    return (int)pair_mask;
}

int dspi_spdif_block_pos(dspi_ctx *c, int32_t set) {
    if (!c || set >= 192) return DSPI_E_INVAL;
    if (set >= 0) c->spdif_pos = (uint32_t)set;
    return (int)c->spdif_pos;
}

int dspi_sync(dspi_ctx *c) {
    if (!c) return DSPI_E_INVAL;
    if (c->device == DSPI_DEVICE_NONE) return DSPI_E_NODEVICE;
    HIPCK(c, hipStreamSynchronize(c->hs));
    return DSPI_OK;
}

int dspi_process(dspi_ctx *c, const void *pcm_in, int bit_depth, uint32_t n_blocks, uint32_t block_len, const dspi_out *out, uint32_t flags) {
    if (!c || !pcm_in || !out) return DSPI_E_INVAL;
    // undefined flag bits are refused, not ignored: a later ABI may give them a meaning that reads further members of dspi_out
    constexpr uint32_t kKnownFlags = DSPI_MEM_DEVICE | DSPI_OUT_TILED | DSPI_OUT_ENABLED_ONLY | DSPI_OUT_I2S_SLOTS | DSPI_OUT_SPDIF | DSPI_OUT_CLIP_FLAGS;
    if (flags & ~kKnownFlags) return fail(c, DSPI_E_INVAL, "dspi_process: undefined flag bits");
    if (c->device == DSPI_DEVICE_NONE) return fail(c, DSPI_E_NODEVICE, "host-only context: the HIP path is the only audio path");
    if ((bit_depth != 16 && bit_depth != 24) || n_blocks == 0 || block_len == 0 || block_len > DSPI_MAX_BLOCK_LEN)
        return fail(c, DSPI_E_INVAL, "bit_depth must be 16/24, 1 <= block_len <= 192, n_blocks >= 1");
    const auto call_t0 = std::chrono::steady_clock::now();
    HIPCK(c, hipSetDevice(c->device));
    int rc = commit_params(c);
    if (rc) return rc;
    // alpha^count (leveller.c:200) on the device = step 1 of include/dspi_detmath.h's powf + its exception table; proven equal to the exact form
    // for the firmware's alphas and every block length by tools/gen_detmath_tables.c — and checked here for the pairs this context really uses
    if (c->lv_checked_count != block_len || c->lv_checked_n != c->lv_alphas.size()) {
        for (const uint32_t bits : c->lv_alphas) {
            float al; memcpy(&al, &bits, 4);
            const float t = dspi_det_powf_tab(al, (float)block_len), e = dspi_det_powf(al, (float)block_len);
            if (memcmp(&t, &e, 4) != 0) return fail(c, DSPI_E_UNSUPPORTED, "leveller alpha^count: this (alpha, block length) is not covered by the device's exception table (include/dspi_detmath_tables.h)");
        }
        c->lv_checked_count = block_len; c->lv_checked_n = c->lv_alphas.size();
    }

    const size_t frames = (size_t)n_blocks * block_len;
    const size_t in_b = (size_t)c->n_streams * frames * (bit_depth == 24 ? 6 : 4);
    const bool tiled = flags & DSPI_OUT_TILED;
    const bool spdif = flags & DSPI_OUT_SPDIF;
    bool spdif_two_pass = false;
    if (spdif) {
        if (tiled || (flags & DSPI_OUT_I2S_SLOTS)) return fail(c, DSPI_E_INVAL, "DSPI_OUT_SPDIF goes with neither DSPI_OUT_TILED nor DSPI_OUT_I2S_SLOTS");
        // The latency layout's output waves encode the subframes themselves.  A launch with lanes on any other kernel (the flag is a
        // property of the output, not of the stream count) runs the chain into a scratch buffer of pair words, row chunk by row chunk,
        // and the subframe encoder from there into `pairs`: the same words, the block position carried the same way.
        bool all_latency = c->flavor != 0;
        for (int lev = 0; lev < 2 && all_latency; lev++)
            for (int k = 1; k <= 4; k++) if (!c->launch_items[lev][k].empty()) all_latency = false;
        spdif_two_pass = !all_latency && out->pairs != nullptr;
    }
    const size_t padded = (size_t)c->n_wg * c->sm.row;          // tiled buffers cover whole tiles
    const size_t pairs_b = tiled ? padded * (c->sm.n_out - 1) * frames * 4 : (size_t)c->n_streams * c->sm.n_pairs * frames * (spdif ? 16 : 8);
    const size_t sub_b = (tiled ? padded : (size_t)c->n_streams) * frames * 4;
    const size_t peaks_b = (size_t)c->n_streams * n_blocks * c->sm.n_ch * 2;
    const bool dev = flags & DSPI_MEM_DEVICE;
    // DSPI_OUT_CLIP_FLAGS: the caller's dspi_out has the ABI-7 member `clip_flags`
    uint16_t *const clip_out = (flags & DSPI_OUT_CLIP_FLAGS) ? out->clip_flags : nullptr;

    KArgs a{};
    a.state = c->d_state; a.dlines = c->d_dlines; a.ring = c->d_ring;
    a.n_streams = c->n_streams; a.n_blocks = n_blocks; a.block_len = block_len; a.bit_depth = (uint32_t)bit_depth;
    a.tiled_out = tiled ? 1u : 0u;
    a.fma = c->fma ? 1u : 0u;
    // (two-pass S/PDIF: the encoder reads the WHOLE scratch chunk, so the chain must write the silent pairs' zero words there as well)
    a.skip_silent = ((flags & DSPI_OUT_ENABLED_ONLY) && !spdif_two_pass) ? 1u : 0u;
    a.i2s_slots = (flags & DSPI_OUT_I2S_SLOTS) ? 1u : 0u;
    if (spdif) { a.spdif = spdif_two_pass ? 0u : 1u; a.spdif_pos = c->spdif_pos; }
    if (c->flavor && !tiled) {      // stream-major layout, packed kernel: the mini lines of the outputs whose rows do not reach the emit wave through their delay line (dspi_chain_pk.inc)
        const size_t xb = (size_t)c->n_wg * 3 * kMaxOut * kChunk * c->sm.row * 4;
        if ((rc = ensure(c, c->d_xwords, c->d_xwords_cap, xb))) return rc;
        a.xwords = c->d_xwords;
    }
    // ---- small calls on host buffers: the drop-in as the firmware's main loop makes it, ONE packet per call (usb_audio_drain_ring,
    // usb_audio.c:1326-1332).  Staged copies would cost four DMA round trips (~100 us) for a few KB; instead the kernels read the packet
    // from, and write their words to, a pinned host area directly (fine-grained, GPU-visible: hipHostMalloc), so the call is two memcpys on
    // the CPU, the launches, and a spin on the stream: its latency is the kernel's. ----
    constexpr size_t kDirectBytes = 2u << 20;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t off_pairs = up(in_b), off_sub = off_pairs + up(out->pairs ? pairs_b : 0), off_peaks = off_sub + up(out->sub ? sub_b : 0),
                 off_clip = off_peaks + up(out->peaks ? peaks_b : 0), direct_b = off_clip + up(clip_out ? (size_t)c->n_streams * 2 : 0);
    const bool direct = !dev && direct_b <= kDirectBytes && !c->no_direct;
    if (direct && direct_b > c->direct_cap) {
        if (c->h_direct) { HIPCK(c, hipStreamSynchronize(c->hs)); (void)hipHostFree(c->h_direct); c->h_direct = nullptr; c->direct_cap = 0; }
        const size_t want = std::max<size_t>(direct_b, 64u << 10);
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, want, hipHostMallocDefault) != hipSuccess) return fail(c, DSPI_E_NOMEM, "hipHostMalloc failed (direct host area)");
        if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { (void)hipHostFree(hp); return fail(c, DSPI_E_HIP, "hipHostGetDevicePointer failed"); }
        c->h_direct = (char *)hp; c->d_direct = (char *)dp; c->direct_cap = want;
    }
    if (dev) {
        a.pcm = pcm_in; a.pairs = out->pairs; a.sub = out->sub; a.peaks = out->peaks;
    } else if (direct) {
        memcpy(c->h_direct, pcm_in, in_b);
        a.pcm = c->d_direct;
        if (out->pairs) a.pairs = reinterpret_cast<int32_t *>(c->d_direct + off_pairs);
        if (out->sub) a.sub = reinterpret_cast<int32_t *>(c->d_direct + off_sub);
        if (out->peaks) a.peaks = reinterpret_cast<uint16_t *>(c->d_direct + off_peaks);
        if (flags & DSPI_OUT_ENABLED_ONLY) {      // (silent parts stay unwritten by the kernels: the caller finds zeros, the firmware's own fill)
            if (out->pairs) memset(c->h_direct + off_pairs, 0, pairs_b);
            if (out->sub) memset(c->h_direct + off_sub, 0, sub_b);
        }
    } else {
        if ((rc = ensure(c, c->d_in, c->d_in_cap, in_b))) return rc;
        a.pcm = c->d_in;
        if (out->pairs) { if ((rc = ensure(c, c->d_pairs, c->d_pairs_cap, pairs_b))) return rc; a.pairs = c->d_pairs; }
        if (out->sub) { if ((rc = ensure(c, c->d_sub, c->d_sub_cap, sub_b))) return rc; a.sub = c->d_sub; }
        if (out->peaks) { if ((rc = ensure(c, c->d_peaks, c->d_peaks_cap, peaks_b))) return rc; a.peaks = c->d_peaks; }
        if (clip_out && (rc = ensure(c, c->d_clip, c->d_clip_cap, (size_t)c->n_streams * 2))) return rc;
        // DSPI_OUT_ENABLED_ONLY leaves the silent parts of pairs / sub unwritten: the staging buffers are copied back whole, so what the
        // caller finds there is zeros (the firmware's own fill), not stale staging memory
        if (flags & DSPI_OUT_ENABLED_ONLY) {
            if (a.pairs) HIPCK(c, hipMemsetAsync(a.pairs, 0, pairs_b, c->hs));
            if (a.sub) HIPCK(c, hipMemsetAsync(a.sub, 0, sub_b, c->hs));
        }
    }
    int32_t *const final_pairs = a.pairs;      // where the caller's pair words / subframes end up (device side)
    uint32_t two_pass_rows = 0;
    if (spdif_two_pass) {
        // scratch for the chain's pair words of a row chunk, capped by BYTES: ~1 GiB worth of rows; one workgroup per CU (256 rows) only
        // while that stays within 2 GiB; never less than one row
        const size_t row_b = (size_t)c->sm.row * c->sm.n_pairs * frames * 8;
        size_t rows = std::max<size_t>(1, ((size_t)1 << 30) / row_b);
        if (rows < 256 && 256 * row_b <= ((size_t)2 << 30)) rows = 256;
        two_pass_rows = (uint32_t)std::min<size_t>(c->n_wg, rows);
        if ((rc = ensure(c, c->d_spdif_words, c->d_spdif_words_cap, (size_t)two_pass_rows * row_b))) return rc;
    }
    a.img = c->d_images;
    a.stream_image = c->d_stream_image;
    // float flavour: lanes with both streams on one image go to the packed kernel (list 1), every other lane to the
    // per-lane-parameter kernel for both lane components in one launch (list 2).  Q28: rows with one image run workgroup-uniform
    // (list 0), rows with several in per-lane-parameter mode (list 2).  A launch covers every image.
    struct Launch { int list; int packed; };
    static const Launch kF32[] = {{1, 1}, {5, 5}, {6, 6}, {7, 7}, {8, 8}, {3, 3}, {4, 4}, {2, 2}};      // lists 3 / 4: per-lane-value rows (packed kernel + value tiles); 5 - 8: latency layout
    static const Launch kQ28[] = {{0, 0}, {2, 2}};
    const Launch *ls = c->flavor ? kF32 : kQ28;
    const int nl = c->flavor ? 8 : 2;
    a.vals = c->d_vals;
    // the chain launches for the rows [r0, r1) (the lists are sorted by row)
    auto launch_rows_1 = [&](uint32_t r0, uint32_t r1) -> int {
        for (int lev = 0; lev < 2; lev++)
            for (int l = 0; l < nl; l++) {
                const auto &items = c->launch_items[lev][ls[l].list];
                if (items.empty()) continue;
                auto by_row = [](const WgItem &it, uint32_t r) { return it.wg < r; };
                const size_t lo = (size_t)(std::lower_bound(items.begin(), items.end(), r0, by_row) - items.begin());
                const size_t hi = (size_t)(std::lower_bound(items.begin(), items.end(), r1, by_row) - items.begin());
                if (hi == lo) continue;
                a.items = c->d_litems + c->launch_item_offset[lev][ls[l].list] + lo;
                hipError_t e = launch_chain(c->flavor, ls[l].packed, lev != 0, a, (uint32_t)(hi - lo), c->hs);
                if (e == hipErrorNotSupported) return fail(c, DSPI_E_UNSUPPORTED, "this flavour has no HIP kernel yet");
                if (e != hipSuccess) return fail(c, DSPI_E_HIP, std::string("chain kernel launch: ") + hipGetErrorString(e));
                c->audio_started = true;      // only now: a call refused for its arguments, or one that could not allocate, leaves a booting device booting (dspi_load_flash_dump)
            }
        return 0;
    };
    const uint32_t row_ = (uint32_t)c->sm.row;
    auto launch_rows = [&](uint32_t r0, uint32_t r1) -> int {
        if (!spdif_two_pass) return launch_rows_1(r0, r1);
        const size_t per_stream = (size_t)c->sm.n_pairs * frames;      // frames of pair words per stream
        for (uint32_t q0 = r0; q0 < r1; q0 += two_pass_rows) {
            const uint32_t q1 = std::min(r1, q0 + two_pass_rows);
            const size_t s0 = (size_t)q0 * row_, s1 = std::min((size_t)q1 * row_, (size_t)c->n_streams);
            a.pairs = c->d_spdif_words; a.pairs_stream0 = (uint32_t)s0;      // the kernels index by absolute stream: stream s0 lands at the scratch's start
            int r = launch_rows_1(q0, q1);
            if (r) return r;
            hipError_t e = launch_spdif(false, c->d_spdif_words, reinterpret_cast<uint32_t *>(final_pairs) + s0 * per_stream * 4, (uint32_t)(s1 - s0), (uint32_t)c->sm.n_pairs,
                                        (uint32_t)frames, row_, q1 - q0, c->spdif_pos, 0u, SpdifRates{c->d_images, c->d_stream_image, (uint32_t)s0}, c->hs);
            if (e != hipSuccess) return fail(c, DSPI_E_HIP, std::string("spdif encoder launch: ") + hipGetErrorString(e));
        }
        return 0;
    };
    // sticky clip flags of every stream (global_status.clip_flags, usb_audio.c:2427-2443), after the chain on the same stream
    auto gather_clip = [&](uint16_t *dst) -> int {
        hipError_t e = launch_clip_gather(c->d_state, c->n_streams, row_, (uint32_t)c->sm.n_slots, (uint32_t)c->sm.clip, dst, c->hs);
        return e == hipSuccess ? 0 : fail(c, DSPI_E_HIP, std::string("clip gather launch: ") + hipGetErrorString(e));
    };
    if (dev) {
        if ((rc = launch_rows(0, c->n_wg))) return rc;
        if (clip_out && (rc = gather_clip(clip_out))) return rc;
        if (spdif) c->spdif_pos = (uint32_t)((c->spdif_pos + frames) % 192u);      // only once everything is enqueued: a failed call leaves the block position alone
        return DSPI_OK;
    }
    if (direct) {
        if ((rc = launch_rows(0, c->n_wg))) return rc;
        if (clip_out && (rc = gather_clip(reinterpret_cast<uint16_t *>(c->d_direct + off_clip)))) return rc;
        // the launches take tens of microseconds: polling answers within a microsecond of their end, a blocking wait adds a wake-up — but only
        // for as long as such launches can take: past the budget the call falls back to the blocking wait (a hung queue, or a host running
        // many contexts, must not pin a core).  The budget: the audio time the call carries (a caller in the firmware's rhythm has exactly
        // that long per call), never less than 300 us, never more than 50 ms; DSPI_DIRECT_SPIN_US (read at dspi_create) overrides it.
        // (The rare 0.5-10 ms calls — BENCH_r05 had one — are not this loop's: its clock check does not fire during them, the thread is off
        //  its core; profiles/r06_realtime_polling.md.)
        // What is polled (round 6): a word in pinned host memory that the stream itself sets to this call's sequence number behind the launches
        // (hipStreamWriteValue32) — a load per poll, no call into the runtime while waiting; hipStreamQuery where that is not available
        // (DSPI_DIRECT_POLL=query forces it).
        if (c->direct_flag < 0) {
            const char *e = getenv("DSPI_DIRECT_POLL");
            c->direct_flag = 0;
            if (!(e && !strcmp(e, "query"))) {
                void *hp = nullptr, *dp = nullptr;
                if (hipHostMalloc(&hp, 64, hipHostMallocDefault) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
                    c->h_done = (uint32_t *)hp; c->d_done = (uint32_t *)dp; *c->h_done = 0u; c->direct_flag = 1;
                } else { if (hp) (void)hipHostFree(hp); (void)hipGetLastError(); }
            }
        }
        bool flagged = false;
        if (c->direct_flag == 1) {
            ++c->direct_seq;
            if (hipStreamWriteValue32(c->hs, c->d_done, c->direct_seq, 0) == hipSuccess) flagged = true;
            else { (void)hipGetLastError(); c->direct_flag = 0; }      // (this runtime / device cannot: the stream is polled from here on)
        }
        hipError_t q = hipSuccess;
        const auto spin_t0 = std::chrono::steady_clock::now();
        const uint64_t audio_us = (uint64_t)frames * 1000000u / 44100u;      // (the slowest rate the firmware runs: an upper bound of the packet's time)
        const auto budget = std::chrono::microseconds(c->direct_spin_us ? (uint64_t)c->direct_spin_us : std::min<uint64_t>(50000u, std::max<uint64_t>(300u, audio_us)));
        uint32_t polls = 0;
        bool fell_back = false;
        if (flagged) {
            volatile const uint32_t *done = c->h_done;
            const uint32_t want = c->direct_seq;
            while (*done != want) {
                __builtin_ia32_pause();
                if ((++polls & 1023u) == 0 && std::chrono::steady_clock::now() - spin_t0 > budget) { q = hipStreamSynchronize(c->hs); fell_back = true; break; }
            }
            std::atomic_thread_fence(std::memory_order_acquire);      // the words the kernels wrote are read after the flag
        } else {
            while ((q = hipStreamQuery(c->hs)) == hipErrorNotReady) {
                if ((++polls & 63u) == 0 && std::chrono::steady_clock::now() - spin_t0 > budget) { q = hipStreamSynchronize(c->hs); fell_back = true; break; }
            }
        }
        {
            const auto t_end = std::chrono::steady_clock::now();
            const uint64_t enq = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(spin_t0 - call_t0).count();
            const uint64_t wait = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_end - spin_t0).count();
            c->direct_stats[0]++;
            if (c->direct_stats[0] > 8) {      // (the context's first calls allocate the pinned area, build the launch lists, load the code objects: not the steady state)
                c->direct_stats[1] += fell_back ? 1u : 0u;
                c->direct_stats[2] = std::max(c->direct_stats[2], enq); c->direct_stats[3] = std::max(c->direct_stats[3], wait);
            }
            c->direct_stats[4] = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(budget).count();
        }
        if (q != hipSuccess) return fail(c, DSPI_E_HIP, std::string("stream: ") + hipGetErrorString(q));
        if (out->pairs) memcpy(out->pairs, c->h_direct + off_pairs, pairs_b);
        if (out->sub) memcpy(out->sub, c->h_direct + off_sub, sub_b);
        if (out->peaks) memcpy(out->peaks, c->h_direct + off_peaks, peaks_b);
        if (clip_out) memcpy(clip_out, c->h_direct + off_clip, (size_t)c->n_streams * 2);
        if (spdif) c->spdif_pos = (uint32_t)((c->spdif_pos + frames) % 192u);
        return DSPI_OK;
    }

    // ---- host buffers (the caller of usb_audio.c:1326-1332 is a host feeding packets): staged through device buffers.  The link moves
    // 4 + 36 bytes per frame, the chain 100 times that, so the call is link-bound; what can be saved is the serialisation: the rows are
    // cut into chunks and chunk i's D2H runs while chunk i+1 computes and chunk i+2 uploads (three streams, events).  The caller's
    // buffers are pinned for the duration of the call (hipHostRegister: ~8 ms per GiB) so that the copies are asynchronous DMA; when
    // that is refused (already registered, read-only mapping ...) the copies still work, just synchronously. ----
    const uint32_t row = (uint32_t)c->sm.row;
    const size_t out_b = (out->pairs ? pairs_b : 0) + (out->sub ? sub_b : 0) + (out->peaks ? peaks_b : 0);
    uint32_t n_chunks = 1;
    if (out_b + in_b >= (32u << 20) && c->n_wg >= 2) {
        n_chunks = (uint32_t)std::min<size_t>({(size_t)8, (size_t)c->n_wg, (out_b + in_b) / (16u << 20)});
        if (n_chunks < 1) n_chunks = 1;
    }
    const uint32_t rows_per = (c->n_wg + n_chunks - 1) / n_chunks;
    n_chunks = (c->n_wg + rows_per - 1) / rows_per;
    struct Pin { void *p; bool on; };
    Pin pins[4] = {{const_cast<void *>(pcm_in), false}, {out->pairs, false}, {out->sub, false}, {out->peaks, false}};
    const size_t pin_b[4] = {in_b, pairs_b, sub_b, peaks_b};
    if (n_chunks > 1)
        for (int i = 0; i < 4; i++)
            if (pins[i].p && pin_b[i] >= (1u << 20)) {
                pins[i].on = hipHostRegister(pins[i].p, pin_b[i], hipHostRegisterDefault) == hipSuccess;
                if (!pins[i].on) (void)hipGetLastError();
            }
    // (error paths too: no copy or kernel may still be in flight on memory that is about to be unregistered)
    auto unpin = [&]() {
        if (c->hs_in) (void)hipStreamSynchronize(c->hs_in);
        (void)hipStreamSynchronize(c->hs);
        if (c->hs_out) (void)hipStreamSynchronize(c->hs_out);
        for (auto &pn : pins) if (pn.on) { (void)hipHostUnregister(pn.p); pn.on = false; }
    };
    if (n_chunks > 1 && !c->hs_in) {
        if (hipStreamCreateWithFlags(&c->hs_in, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c->hs_out, hipStreamNonBlocking) != hipSuccess) {
            unpin(); return fail(c, DSPI_E_HIP, "stream creation for the host-buffer pipeline failed");
        }
    }
    while (n_chunks > 1 && c->pipe_events.size() < 2 * (size_t)n_chunks) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { unpin(); return fail(c, DSPI_E_HIP, "event creation failed"); }
        c->pipe_events.push_back(e);
    }
    const size_t bpf = bit_depth == 24 ? 6 : 4;
    auto fail_hip = [&](hipError_t e, const char *what) { unpin(); return fail(c, DSPI_E_HIP, std::string(what) + ": " + hipGetErrorString(e)); };
    hipError_t he;
    for (uint32_t ch = 0; ch < n_chunks; ch++) {
        const uint32_t r0 = ch * rows_per, r1 = std::min(c->n_wg, r0 + rows_per);
        const size_t s0 = (size_t)r0 * row, s1 = std::min((size_t)r1 * row, (size_t)c->n_streams);      // streams of the chunk
        hipStream_t sin = n_chunks > 1 ? c->hs_in : c->hs, sout = n_chunks > 1 ? c->hs_out : c->hs;
        if ((he = hipMemcpyAsync(static_cast<char *>(c->d_in) + s0 * frames * bpf, static_cast<const char *>(pcm_in) + s0 * frames * bpf, (s1 - s0) * frames * bpf,
                                 hipMemcpyHostToDevice, sin)) != hipSuccess) return fail_hip(he, "H2D");
        if (n_chunks > 1) {
            if ((he = hipEventRecord(c->pipe_events[2 * ch], sin)) != hipSuccess || (he = hipStreamWaitEvent(c->hs, c->pipe_events[2 * ch], 0)) != hipSuccess) return fail_hip(he, "event");
        }
        if ((rc = launch_rows(r0, r1))) { unpin(); return rc; }
        if (n_chunks > 1) {
            if ((he = hipEventRecord(c->pipe_events[2 * ch + 1], c->hs)) != hipSuccess || (he = hipStreamWaitEvent(sout, c->pipe_events[2 * ch + 1], 0)) != hipSuccess) return fail_hip(he, "event");
        }
        // tiled words: whole tiles [tile][...]; stream-major: [stream][...] — either way a row range is one contiguous piece
        const size_t t0 = tiled ? (size_t)r0 * row : s0, t1 = tiled ? (size_t)r1 * row : s1;
        if (out->pairs) {
            const size_t per = tiled ? (size_t)(c->sm.n_out - 1) * frames * 4 : (size_t)c->sm.n_pairs * frames * (spdif ? 16 : 8);
            if ((he = hipMemcpyAsync(reinterpret_cast<char *>(out->pairs) + t0 * per, reinterpret_cast<char *>(final_pairs) + t0 * per, (t1 - t0) * per, hipMemcpyDeviceToHost, sout)) != hipSuccess) return fail_hip(he, "D2H pairs");
        }
        if (out->sub) {
            const size_t per = frames * 4;
            if ((he = hipMemcpyAsync(reinterpret_cast<char *>(out->sub) + t0 * per, reinterpret_cast<char *>(c->d_sub) + t0 * per, (t1 - t0) * per, hipMemcpyDeviceToHost, sout)) != hipSuccess) return fail_hip(he, "D2H sub");
        }
        if (out->peaks) {
            const size_t per = (size_t)n_blocks * c->sm.n_ch * 2;
            if ((he = hipMemcpyAsync(reinterpret_cast<char *>(out->peaks) + s0 * per, reinterpret_cast<char *>(c->d_peaks) + s0 * per, (s1 - s0) * per, hipMemcpyDeviceToHost, sout)) != hipSuccess) return fail_hip(he, "D2H peaks");
        }
    }
    if (clip_out) {
        if ((rc = gather_clip(c->d_clip))) { unpin(); return rc; }
        if ((he = hipMemcpyAsync(clip_out, c->d_clip, (size_t)c->n_streams * 2, hipMemcpyDeviceToHost, c->hs)) != hipSuccess) return fail_hip(he, "D2H clip flags");
    }
    if (n_chunks > 1) {
        if ((he = hipStreamSynchronize(c->hs_out)) != hipSuccess) return fail_hip(he, "sync");
        if ((he = hipStreamSynchronize(c->hs_in)) != hipSuccess) return fail_hip(he, "sync");
    }
    if ((he = hipStreamSynchronize(c->hs)) != hipSuccess) return fail_hip(he, "sync");
    unpin();
    if (spdif) c->spdif_pos = (uint32_t)((c->spdif_pos + frames) % 192u);
    return DSPI_OK;
}

}  // extern "C"
