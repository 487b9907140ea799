// evcap_api.cpp -- the C ABI declared in include/evcap.h: MP4 samples -> NAL units -> decoder -> frames in presentation
// order -> BGR as cv2.VideoCapture hands it over (evenvizion_component.py:132, video_processing.py:58,70).
#include "../../include/evcap.h"

#include <cstdio>
#include <deque>
#include <mutex>

#include "evc_h264.h"
#include "evc_mp4.h"

using namespace evc;

struct evcap {
    std::vector<uint8_t> file;
    Mp4Track track;
    Decoder dec;
    size_t next_sample = 0;
    bool flushed = false;
    std::deque<PicPtr> ready;
    PicPtr last;
    int bgr_mode = EVCAP_BGR_SWSCALE_X86;
    int width = 0, height = 0, crop_l = 0, crop_t = 0;
    // presentation window of the edit list in media time-scale units (only when the track has exactly one edit)
    bool honour_edits = true, have_window = false;
    int64_t win_start = 0, win_end = 0;
    int dropped_by_edit = 0;
    bool started = false;
    std::string err;
};

static std::string g_open_error;
static std::mutex g_open_mutex;

namespace {

int classify(const std::string& msg) {
    if (msg.rfind("mp4:", 0) == 0 || msg.rfind("sps:", 0) == 0 || msg.rfind("pps:", 0) == 0) return EVCAP_ERR_FORMAT;
    return EVCAP_ERR_STREAM;
}

void feed_nals(evcap* c, const uint8_t* d, size_t n) {
    const int ls = c->track.nal_length_size;
    size_t o = 0;
    while (o + ls <= n) {
        size_t len = 0;
        for (int i = 0; i < ls; ++i) len = (len << 8) | d[o + i];
        o += ls;
        if (len > n - o) fail("mp4: a NAL unit of %zu bytes overruns its sample", len);
        if (len) c->dec.decode_nal(d + o, len);
        o += len;
    }
}

// decodes until at least one picture is ready or the stream ends
void pump(evcap* c) {
    while (c->ready.empty() && !c->flushed) {
        if (c->next_sample < c->track.samples.size()) {
            const Mp4Sample& s = c->track.samples[c->next_sample++];
            feed_nals(c, c->file.data() + s.offset, s.size);
        } else {
            c->dec.flush();
            c->flushed = true;
        }
        for (auto& p : c->dec.take_output()) c->ready.push_back(p);
    }
}

int open_common(evcap* c, evcap** out) {
    try {
        c->track = mp4_parse(c->file);
        // ISO/IEC 14496-12 8.6.6: an edit maps [media_time, media_time + segment_duration) onto the presentation; samples
        // whose composition time lies outside it are decoded (they may be references) but not presented.  FFmpeg's mov
        // demuxer, which cv2.VideoCapture reads through, does the same (it flags such samples "discard"), and that is why
        // the reference's run of test_video.mp4 saw 121 of its 122 samples: the last one is composed 507 ticks after the
        // end of the only edit.  segment_duration is in movie time-scale units; rounded to nearest like av_rescale.
        if (c->track.edits.size() == 1 && c->track.edits[0].media_time >= 0 && c->track.edits[0].segment_duration > 0 &&
            c->track.movie_timescale > 0) {
            const Mp4Track::Edit& e = c->track.edits[0];
            const unsigned __int128 num = (unsigned __int128)e.segment_duration * c->track.timescale;
            const uint64_t dur = (uint64_t)((num + c->track.movie_timescale / 2) / c->track.movie_timescale);
            c->have_window = true;
            c->win_start = e.media_time;
            c->win_end = e.media_time + (int64_t)dur;
        }
        for (auto& s : c->track.sps) c->dec.decode_nal(s.data(), s.size());
        for (auto& p : c->track.pps) c->dec.decode_nal(p.data(), p.size());
        // decode ahead to the first picture so that the geometry is known (in-band parameter sets may override avcC)
        pump(c);
        const SPS* sps = c->dec.active_sps();
        if (!sps) fail("mp4: the video track holds no decodable picture");
        c->width = sps->width();
        c->height = sps->height();
        c->crop_l = sps->crop_l;
        c->crop_t = sps->crop_t;
    } catch (const std::exception& e) {
        std::lock_guard<std::mutex> g(g_open_mutex);
        g_open_error = e.what();
        int rc = classify(g_open_error);
        delete c;
        *out = nullptr;
        return rc;
    }
    *out = c;
    return EVCAP_OK;
}

inline uint8_t sat8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
inline int16_t sat16(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

// libswscale yuv2rgb, ITU-R BT.601 coefficients {104597, 132201, 25675, 53279} / 65536, limited range
struct SwsX86 {
    // 13-bit coefficients: round(c * 8192 / 65536) as roundToInt16 computes them
    static constexpr int yc = 9539, vr = 13075, ub = 16525, ug = -3209, vg = -6660;
    static inline void px(int y, int u, int v, uint8_t* bgr) {
        int16_t y8 = (int16_t)((y << 3) - 128), u8 = sat16((u << 3) - 1024), v8 = sat16((v << 3) - 1024);
        int Y = (y8 * yc) >> 16;  // pmulhw
        int UB = (u8 * ub) >> 16, VR = (v8 * vr) >> 16, CG = sat16(((u8 * ug) >> 16) + ((v8 * vg) >> 16));
        bgr[0] = sat8(sat16(Y + UB));
        bgr[1] = sat8(sat16(Y + CG));
        bgr[2] = sat8(sat16(Y + VR));
    }
};

struct SwsC {
    long long cy, crv, cbu, cgu, cgv;
    SwsC() {
        cy = ((1LL << 16) * 255) / 219;
        crv = ((104597LL << 16) + 0x8000) / cy;
        cbu = ((132201LL << 16) + 0x8000) / cy;
        cgu = ((-25675LL * (1 << 16)) + 0x8000) / cy;
        cgv = ((-53279LL * (1 << 16)) + 0x8000) / cy;
    }
    inline int ytab(long long idx) const {  // y_table[384 + idx]
        long long yb = -(16LL << 16) + idx * cy;
        return sat8((int)((yb + 0x8000) >> 16));
    }
    inline void px(int y, int u, int v, uint8_t* bgr) const {
        long long dr = ((v * crv) >> 16) - (crv >> 9);
        long long db = ((u * cbu) >> 16) - (cbu >> 9);
        long long dg = ((u * cgu) >> 16) - (cgu >> 9) + ((v * cgv) >> 16) - (cgv >> 9);
        bgr[0] = (uint8_t)ytab(y + db);
        bgr[1] = (uint8_t)ytab(y + dg);
        bgr[2] = (uint8_t)ytab(y + dr);
    }
};

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int evcap_open(const char* path, evcap** out) {
    if (!out) return EVCAP_ERR_INVALID;
    *out = nullptr;
    if (!path) return EVCAP_ERR_INVALID;
    FILE* f = std::fopen(path, "rb");
    if (!f) {
        std::lock_guard<std::mutex> g(g_open_mutex);
        g_open_error = std::string("cannot open ") + path;
        return EVCAP_ERR_IO;
    }
    evcap* c = new evcap;
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz < 0) sz = 0;
    c->file.resize((size_t)sz);
    size_t got = sz ? std::fread(c->file.data(), 1, (size_t)sz, f) : 0;
    std::fclose(f);
    if (got != (size_t)sz) {
        delete c;
        std::lock_guard<std::mutex> g(g_open_mutex);
        g_open_error = std::string("short read on ") + path;
        return EVCAP_ERR_IO;
    }
    return open_common(c, out);
}

int evcap_open_memory(const uint8_t* data, uint64_t size, evcap** out) {
    if (!out) return EVCAP_ERR_INVALID;
    *out = nullptr;
    if (!data && size) return EVCAP_ERR_INVALID;
    evcap* c = new evcap;
    c->file.assign(data, data + size);
    return open_common(c, out);
}

void evcap_close(evcap* c) { delete c; }

int evcap_info(evcap* c, int* width, int* height, int* frame_count, double* fps) {
    if (!c) return EVCAP_ERR_INVALID;
    if (width) *width = c->width;
    if (height) *height = c->height;
    if (frame_count) *frame_count = (int)c->track.samples.size();
    if (fps) {
        double f = 0;
        size_t n = c->track.samples.size();
        if (n > 1 && c->track.timescale) {
            double span = double(c->track.samples[n - 1].dts - c->track.samples[0].dts);
            if (span > 0) f = (n - 1) * double(c->track.timescale) / span;
        }
        *fps = f;
    }
    return EVCAP_OK;
}

int evcap_set_honour_edit_list(evcap* c, int on) {
    if (!c) return EVCAP_ERR_INVALID;
    if (c->started) return EVCAP_ERR_INVALID;  // only before the first read
    c->honour_edits = on != 0;
    return EVCAP_OK;
}

int evcap_set_bgr_mode(evcap* c, int mode) {
    if (!c || (mode != EVCAP_BGR_SWSCALE_X86 && mode != EVCAP_BGR_SWSCALE_C)) return EVCAP_ERR_INVALID;
    c->bgr_mode = mode;
    return EVCAP_OK;
}

static int next_picture(evcap* c, PicPtr& p) {
    for (;;) {
        try {
            pump(c);
        } catch (const std::exception& e) {
            c->err = e.what();
            c->flushed = true;
            c->ready.clear();
            return classify(c->err);
        }
        if (c->ready.empty()) return EVCAP_EOF;
        p = c->ready.front();
        c->ready.pop_front();
        // one picture per sample: Picture::id counts pictures in decoding order = sample index
        if (c->honour_edits && c->have_window && (size_t)p->id < c->track.samples.size()) {
            const int64_t cts = c->track.samples[(size_t)p->id].pts;
            if (cts < c->win_start || cts >= c->win_end) {
                ++c->dropped_by_edit;
                continue;
            }
        }
        c->last = p;
        c->started = true;
        return EVCAP_OK;
    }
}

int evcap_read_bgr(evcap* c, uint8_t* dst, int64_t stride) {
    if (!c || !dst || stride < (int64_t)c->width * 3) return EVCAP_ERR_INVALID;
    PicPtr p;
    int rc = next_picture(c, p);
    if (rc != EVCAP_OK) return rc;
    static const SwsC swsc;
    if (c->bgr_mode == EVCAP_BGR_SWSCALE_X86) {
        // the same arithmetic through tables: every term of SwsX86::px depends on one sample only (luma, Cb or Cr), the sums of
        // two terms stay far inside int16 (|Y| < 300, |UB|, |VR|, |CG| < 300), so sat16(Y + T) == Y + T and only the final
        // saturation to a byte remains.  (5.2 -> 0.6 ms per 1170 x 658 frame.)
        static int16_t TY[256], TUB[256], TVR[256], TUG[256], TVG[256];
        static const bool init = [] {
            for (int i = 0; i < 256; ++i) {
                const int16_t y8 = (int16_t)((i << 3) - 128), c8 = sat16((i << 3) - 1024);
                TY[i] = (int16_t)((y8 * SwsX86::yc) >> 16);
                TUB[i] = (int16_t)((c8 * SwsX86::ub) >> 16); TVR[i] = (int16_t)((c8 * SwsX86::vr) >> 16);
                TUG[i] = (int16_t)((c8 * SwsX86::ug) >> 16); TVG[i] = (int16_t)((c8 * SwsX86::vg) >> 16);
            }
            return true;
        }();
        (void)init;
        for (int y = 0; y < c->height; ++y) {
            const int sy = y + c->crop_t;
            const uint8_t* Y = &p->Y[(size_t)sy * p->stride + c->crop_l];
            const uint8_t* U = &p->Cb[(size_t)(sy >> 1) * p->cstride];
            const uint8_t* V = &p->Cr[(size_t)(sy >> 1) * p->cstride];
            uint8_t* o = dst + (size_t)y * stride;
            int x = 0;
            if (!(c->crop_l & 1)) {                          // two luma samples share a chroma sample
                const uint8_t* Uc = U + (c->crop_l >> 1);
                const uint8_t* Vc = V + (c->crop_l >> 1);
                for (; x + 1 < c->width; x += 2) {
                    const int u = Uc[x >> 1], v = Vc[x >> 1];
                    const int ub = TUB[u], vr = TVR[v], cg = sat16(TUG[u] + TVG[v]);
                    const int y0 = TY[Y[x]], y1 = TY[Y[x + 1]];
                    uint8_t* q = o + 3 * x;
                    q[0] = sat8(y0 + ub); q[1] = sat8(y0 + cg); q[2] = sat8(y0 + vr);
                    q[3] = sat8(y1 + ub); q[4] = sat8(y1 + cg); q[5] = sat8(y1 + vr);
                }
            }
            for (; x < c->width; ++x) {
                const int cx = (x + c->crop_l) >> 1;
                const int u = U[cx], v = V[cx], yy = TY[Y[x]];
                const int cg = sat16(TUG[u] + TVG[v]);
                o[3 * x] = sat8(yy + TUB[u]);
                o[3 * x + 1] = sat8(yy + cg);
                o[3 * x + 2] = sat8(yy + TVR[v]);
            }
        }
        return EVCAP_OK;
    }
    for (int y = 0; y < c->height; ++y) {
        const int sy = y + c->crop_t;
        const uint8_t* Y = &p->Y[(size_t)sy * p->stride + c->crop_l];
        const uint8_t* U = &p->Cb[(size_t)(sy >> 1) * p->cstride];
        const uint8_t* V = &p->Cr[(size_t)(sy >> 1) * p->cstride];
        uint8_t* o = dst + (size_t)y * stride;
        for (int x = 0; x < c->width; ++x) {
            int cx = (x + c->crop_l) >> 1;
            swsc.px(Y[x], U[cx], V[cx], o + 3 * x);
        }
    }
    return EVCAP_OK;
}

int evcap_read_yuv420(evcap* c, uint8_t* y, int64_t ystride, uint8_t* cb, uint8_t* cr, int64_t cstride) {
    if (!c || !y || !cb || !cr || ystride < c->width || cstride < (c->width + 1) / 2) return EVCAP_ERR_INVALID;
    if ((c->crop_l | c->crop_t) & 1) return EVCAP_ERR_INVALID;
    PicPtr p;
    int rc = next_picture(c, p);
    if (rc != EVCAP_OK) return rc;
    for (int r = 0; r < c->height; ++r) std::memcpy(y + (size_t)r * ystride, &p->Y[(size_t)(r + c->crop_t) * p->stride + c->crop_l], (size_t)c->width);
    const int cw = (c->width + 1) / 2, ch = (c->height + 1) / 2;
    for (int r = 0; r < ch; ++r) {
        std::memcpy(cb + (size_t)r * cstride, &p->Cb[(size_t)(r + c->crop_t / 2) * p->cstride + c->crop_l / 2], (size_t)cw);
        std::memcpy(cr + (size_t)r * cstride, &p->Cr[(size_t)(r + c->crop_t / 2) * p->cstride + c->crop_l / 2], (size_t)cw);
    }
    return EVCAP_OK;
}

int evcap_last_frame_info(evcap* c, int* poc, int* decode_index, int* slice_type) {
    if (!c || !c->last) return EVCAP_ERR_INVALID;
    if (poc) *poc = c->last->poc;
    if (decode_index) *decode_index = c->last->id;
    if (slice_type) *slice_type = c->last->slice_type_first;
    return EVCAP_OK;
}

int evcap_stats(evcap* c, int64_t* out, int n) {
    if (!c) return EVCAP_ERR_INVALID;
    const Stats& s = c->dec.stats();
    const int64_t v[] = {s.mbs, s.i4, s.i8, s.i16, s.ipcm, s.p_skip, s.b_skip, s.b_direct, s.inter, s.t8x8, s.bipred_blocks,
                         s.explicit_wp_blocks, s.implicit_wp_blocks, s.sub8x8, s.temporal_direct_mbs, s.spatial_direct_mbs, s.mmco_ops,
                         s.list_mods, s.long_term, s.slices[0], s.slices[1], s.slices[2], s.max_ref_idx};
    const int total = (int)(sizeof v / sizeof v[0]);
    for (int i = 0; i < n && i < total && out; ++i) out[i] = v[i];
    return total;
}

const char* evcap_last_error(evcap* c) {
    if (c) return c->err.c_str();
    std::lock_guard<std::mutex> g(g_open_mutex);
    static thread_local std::string copy;
    copy = g_open_error;
    return copy.c_str();
}

#pragma GCC visibility pop
}  // extern "C"
