// evc_mp4.cpp -- see evc_mp4.h.  Box layout per ISO/IEC 14496-12 (moov/trak/mdia/minf/stbl and the sample tables
// stsd/stts/ctts/stsc/stsz/stco/co64/stss) and ISO/IEC 14496-15 (avc1 sample entry, avcC configuration record).
#include "evc_mp4.h"

#include <algorithm>

#include "evc_h264.h"

namespace evc {
namespace {

struct Rd {
    const uint8_t* d;
    uint64_t n;
    uint64_t u(uint64_t o, int bytes) const {
        if (o + bytes > n) fail("mp4: truncated box (offset %llu)", (unsigned long long)o);
        uint64_t v = 0;
        for (int i = 0; i < bytes; ++i) v = (v << 8) | d[o + i];
        return v;
    }
};

struct Box {
    uint32_t type;
    uint64_t body, end;  // payload range
};

constexpr uint32_t fourcc(const char (&s)[5]) {
    return (uint32_t(uint8_t(s[0])) << 24) | (uint32_t(uint8_t(s[1])) << 16) | (uint32_t(uint8_t(s[2])) << 8) |
           uint32_t(uint8_t(s[3]));
}

std::vector<Box> children(const Rd& r, uint64_t o, uint64_t e) {
    std::vector<Box> out;
    while (o + 8 <= e) {
        uint64_t sz = r.u(o, 4);
        uint32_t tp = (uint32_t)r.u(o + 4, 4);
        uint64_t hdr = 8;
        if (sz == 1) {
            sz = r.u(o + 8, 8);
            hdr = 16;
        } else if (sz == 0) {
            sz = e - o;
        }
        if (sz < hdr || o + sz > e) fail("mp4: box size %llu does not fit its parent", (unsigned long long)sz);
        out.push_back({tp, o + hdr, o + sz});
        o += sz;
    }
    return out;
}

const Box* find(const std::vector<Box>& v, uint32_t tp) {
    for (auto& b : v)
        if (b.type == tp) return &b;
    return nullptr;
}

}  // namespace

Mp4Track mp4_parse(const std::vector<uint8_t>& data) {
    Rd r{data.data(), data.size()};
    auto top = children(r, 0, r.n);
    const Box* moov = find(top, fourcc("moov"));
    if (!moov) fail("mp4: no moov box (not an ISO media file, or a fragmented one)");
    auto mv = children(r, moov->body, moov->end);
    uint32_t movie_ts = 0;
    if (const Box* mvhd = find(mv, fourcc("mvhd"))) {
        int ver = (int)r.u(mvhd->body, 1);
        movie_ts = (uint32_t)r.u(mvhd->body + (ver == 1 ? 20 : 12), 4);
    }
    for (auto& tb : mv) {
        if (tb.type != fourcc("trak")) continue;
        auto tk = children(r, tb.body, tb.end);
        const Box* mdia = find(tk, fourcc("mdia"));
        if (!mdia) continue;
        auto md = children(r, mdia->body, mdia->end);
        const Box* hdlr = find(md, fourcc("hdlr"));
        if (!hdlr || r.u(hdlr->body + 8, 4) != fourcc("vide")) continue;
        Mp4Track t;
        t.movie_timescale = movie_ts;
        const Box* mdhd = find(md, fourcc("mdhd"));
        if (!mdhd) fail("mp4: video track without mdhd");
        {
            int ver = (int)r.u(mdhd->body, 1);
            if (ver == 1) {
                t.timescale = (uint32_t)r.u(mdhd->body + 20, 4);
                t.duration = r.u(mdhd->body + 24, 8);
            } else {
                t.timescale = (uint32_t)r.u(mdhd->body + 12, 4);
                t.duration = r.u(mdhd->body + 16, 4);
            }
        }
        if (const Box* edts = find(tk, fourcc("edts"))) {
            auto ed = children(r, edts->body, edts->end);
            if (const Box* elst = find(ed, fourcc("elst"))) {
                int ver = (int)r.u(elst->body, 1);
                uint32_t n = (uint32_t)r.u(elst->body + 4, 4);
                uint64_t o = elst->body + 8;
                for (uint32_t i = 0; i < n; ++i) {
                    Mp4Track::Edit e;
                    if (ver == 1) {
                        e.segment_duration = r.u(o, 8);
                        e.media_time = (int64_t)r.u(o + 8, 8);
                        o += 20;
                    } else {
                        e.segment_duration = r.u(o, 4);
                        e.media_time = (int32_t)r.u(o + 4, 4);
                        o += 12;
                    }
                    t.edits.push_back(e);
                }
            }
        }
        const Box* minf = find(md, fourcc("minf"));
        if (!minf) fail("mp4: video track without minf");
        auto mi = children(r, minf->body, minf->end);
        const Box* stbl = find(mi, fourcc("stbl"));
        if (!stbl) fail("mp4: video track without stbl");
        auto st = children(r, stbl->body, stbl->end);

        // ---- stsd: the first sample entry must be avc1/avc3 with an avcC record
        const Box* stsd = find(st, fourcc("stsd"));
        if (!stsd) fail("mp4: no stsd");
        {
            uint64_t o = stsd->body + 8;
            uint64_t esz = r.u(o, 4);
            uint32_t etp = (uint32_t)r.u(o + 4, 4);
            if (etp != fourcc("avc1") && etp != fourcc("avc3"))
                fail("mp4: video sample entry '%c%c%c%c' is not AVC (only H.264 is decoded)", etp >> 24, (etp >> 16) & 255,
                     (etp >> 8) & 255, etp & 255);
            t.width = (int)r.u(o + 32, 2);
            t.height = (int)r.u(o + 34, 2);
            auto sub = children(r, o + 86, o + esz);
            const Box* avcc = find(sub, fourcc("avcC"));
            if (!avcc) fail("mp4: avc1 sample entry without avcC");
            uint64_t a = avcc->body;
            if (r.u(a, 1) != 1) fail("mp4: avcC configurationVersion != 1");
            t.nal_length_size = int(r.u(a + 4, 1) & 3) + 1;
            int nsps = int(r.u(a + 5, 1) & 31);
            a += 6;
            for (int i = 0; i < nsps; ++i) {
                int len = (int)r.u(a, 2);
                if (a + 2 + len > avcc->end) fail("mp4: avcC SPS overruns the box");
                t.sps.emplace_back(r.d + a + 2, r.d + a + 2 + len);
                a += 2 + len;
            }
            int npps = (int)r.u(a, 1);
            a += 1;
            for (int i = 0; i < npps; ++i) {
                int len = (int)r.u(a, 2);
                if (a + 2 + len > avcc->end) fail("mp4: avcC PPS overruns the box");
                t.pps.emplace_back(r.d + a + 2, r.d + a + 2 + len);
                a += 2 + len;
            }
        }

        // ---- sample sizes
        const Box* stsz = find(st, fourcc("stsz"));
        if (!stsz) fail("mp4: no stsz (stz2 is not supported)");
        uint32_t fixed = (uint32_t)r.u(stsz->body + 4, 4);
        uint32_t count = (uint32_t)r.u(stsz->body + 8, 4);
        if (count > (1u << 24)) fail("mp4: implausible sample count %u", count);
        t.samples.resize(count);
        for (uint32_t i = 0; i < count; ++i) t.samples[i].size = fixed ? fixed : (uint32_t)r.u(stsz->body + 12 + 4ull * i, 4);

        // ---- chunk offsets + sample-to-chunk
        std::vector<uint64_t> chunk_off;
        if (const Box* stco = find(st, fourcc("stco"))) {
            uint32_t n = (uint32_t)r.u(stco->body + 4, 4);
            for (uint32_t i = 0; i < n; ++i) chunk_off.push_back(r.u(stco->body + 8 + 4ull * i, 4));
        } else if (const Box* co64 = find(st, fourcc("co64"))) {
            uint32_t n = (uint32_t)r.u(co64->body + 4, 4);
            for (uint32_t i = 0; i < n; ++i) chunk_off.push_back(r.u(co64->body + 8 + 8ull * i, 8));
        } else {
            fail("mp4: no stco/co64");
        }
        const Box* stsc = find(st, fourcc("stsc"));
        if (!stsc) fail("mp4: no stsc");
        {
            uint32_t n = (uint32_t)r.u(stsc->body + 4, 4);
            struct E {
                uint32_t first, per, desc;
            };
            std::vector<E> es;
            for (uint32_t i = 0; i < n; ++i) {
                uint64_t o = stsc->body + 8 + 12ull * i;
                es.push_back({(uint32_t)r.u(o, 4), (uint32_t)r.u(o + 4, 4), (uint32_t)r.u(o + 8, 4)});
            }
            uint32_t s = 0;
            for (size_t ei = 0; ei < es.size() && s < count; ++ei) {
                uint32_t last = (ei + 1 < es.size()) ? es[ei + 1].first - 1 : (uint32_t)chunk_off.size();
                for (uint32_t c = es[ei].first; c <= last && s < count; ++c) {
                    if (c == 0 || c > chunk_off.size()) fail("mp4: stsc refers to chunk %u of %zu", c, chunk_off.size());
                    uint64_t o = chunk_off[c - 1];
                    for (uint32_t k = 0; k < es[ei].per && s < count; ++k) {
                        t.samples[s].offset = o;
                        o += t.samples[s].size;
                        ++s;
                    }
                }
            }
            if (s != count) fail("mp4: sample tables cover %u of %u samples", s, count);
        }
        for (auto& s : t.samples)
            if (s.offset + s.size > r.n) fail("mp4: a sample lies outside the file");

        // ---- timing
        if (const Box* stts = find(st, fourcc("stts"))) {
            uint32_t n = (uint32_t)r.u(stts->body + 4, 4);
            int64_t dts = 0;
            uint32_t s = 0;
            for (uint32_t i = 0; i < n; ++i) {
                uint32_t cnt = (uint32_t)r.u(stts->body + 8 + 8ull * i, 4), delta = (uint32_t)r.u(stts->body + 12 + 8ull * i, 4);
                for (uint32_t k = 0; k < cnt && s < count; ++k) {
                    t.samples[s++].dts = dts;
                    dts += delta;
                }
            }
        }
        for (auto& s : t.samples) s.pts = s.dts;
        if (const Box* ctts = find(st, fourcc("ctts"))) {
            int ver = (int)r.u(ctts->body, 1);
            uint32_t n = (uint32_t)r.u(ctts->body + 4, 4);
            uint32_t s = 0;
            for (uint32_t i = 0; i < n; ++i) {
                uint32_t cnt = (uint32_t)r.u(ctts->body + 8 + 8ull * i, 4);
                uint32_t raw = (uint32_t)r.u(ctts->body + 12 + 8ull * i, 4);
                int64_t off = ver ? (int64_t)(int32_t)raw : (int64_t)raw;
                for (uint32_t k = 0; k < cnt && s < count; ++k) t.samples[s++].pts += off;
            }
        }
        if (const Box* stss = find(st, fourcc("stss"))) {
            uint32_t n = (uint32_t)r.u(stss->body + 4, 4);
            for (uint32_t i = 0; i < n; ++i) {
                uint32_t k = (uint32_t)r.u(stss->body + 8 + 4ull * i, 4);
                if (k >= 1 && k <= count) t.samples[k - 1].sync = true;
            }
        } else {
            for (auto& s : t.samples) s.sync = true;
        }
        return t;
    }
    fail("mp4: no video track");
}

}  // namespace evc
