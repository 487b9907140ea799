// evc_h264_cabac.cpp -- the CABAC arithmetic decoding engine of ITU-T Rec. H.264 clause 9.3: initialisation of the
// context variables (9.3.1.1), of the decoding engine (9.3.1.2), DecodeDecision (9.3.3.2.1), DecodeBypass (9.3.3.2.3)
// and DecodeTerminate (9.3.3.2.2.3), written as the flowcharts state them (9-bit codIOffset, one bit per renormalisation
// step).
#include "evc_h264_int.h"

namespace evc {

void Cabac::init_engine(const uint8_t* data, const uint8_t* data_end) {
    p = data;
    end = data_end;
    bits_left = 0;
    cur = 0;
    range = 510;
    offset = 0;
    for (int i = 0; i < 9; ++i) offset = (offset << 1) | (uint32_t)read_bit();
    if (offset == 510 || offset == 511) fail("cabac: codIOffset %u after initialisation (illegal bitstream, 9.3.1.2)", offset);
}

void Cabac::init_contexts(int slice_type, int cabac_init_idc, int slice_qp) {
    const int8_t(*tab)[2] = nullptr;
    if (slice_type == SLICE_I) {
        tab = kCabacInitI;
    } else if (cabac_init_idc == 0) {
        tab = kCabacInitPB0;
    } else {
        fail("cabac: cabac_init_idc %d -- only the tables for I slices and cabac_init_idc 0 are carried (x264 always writes 0)",
             cabac_init_idc);
    }
    const int qp = clip3(0, 51, slice_qp);
    for (int i = 0; i < 460; ++i) {
        int pre = clip3(1, 126, ((tab[i][0] * qp) >> 4) + tab[i][1]);
        if (pre <= 63)
            state[i] = (uint8_t)(((63 - pre) << 1) | 0);
        else
            state[i] = (uint8_t)(((pre - 64) << 1) | 1);
    }
    for (int i = 460; i < 1024; ++i) state[i] = 0;
}

int Cabac::decision(int ctx) {
    uint8_t& s = state[ctx];
    int pstate = s >> 1, mps = s & 1;
    uint32_t rlps = kRangeTabLPS[pstate][(range >> 6) & 3];
    range -= rlps;
    int bin;
    if (offset >= range) {
        bin = !mps;
        offset -= range;
        range = rlps;
        if (pstate == 0) mps = 1 - mps;
        pstate = kTransIdxLPS[pstate];
    } else {
        bin = mps;
        if (pstate < 62) ++pstate;
    }
    s = (uint8_t)((pstate << 1) | mps);
    while (range < 256) {
        range <<= 1;
        offset = (offset << 1) | (uint32_t)read_bit();
    }
    return bin;
}

int Cabac::bypass() {
    offset = (offset << 1) | (uint32_t)read_bit();
    if (offset >= range) {
        offset -= range;
        return 1;
    }
    return 0;
}

int Cabac::terminate() {
    range -= 2;
    if (offset >= range) return 1;  // no renormalisation; the next unread bit is the flush's final 1 (rbsp_stop_one_bit)
    while (range < 256) {
        range <<= 1;
        offset = (offset << 1) | (uint32_t)read_bit();
    }
    return 0;
}

}  // namespace evc
