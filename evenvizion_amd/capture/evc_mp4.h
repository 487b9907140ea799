// evc_mp4.h -- ISO base media file (ISO/IEC 14496-12) demultiplexer for the capture source: finds the first AVC video
// track of an .mp4/.mov and lists its samples.  Replaces the container half of cv2.VideoCapture(path)
// (/root/reference/evenvizion/examples/evenvizion_component.py:132).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace evc {

struct Mp4Sample {
    uint64_t offset = 0;
    uint32_t size = 0;
    int64_t dts = 0, pts = 0;  // media time-scale units; pts = dts + composition offset
    bool sync = false;
};

struct Mp4Track {
    uint32_t timescale = 0;
    uint64_t duration = 0;
    int width = 0, height = 0;  // from the sample entry
    int nal_length_size = 4;
    std::vector<std::vector<uint8_t>> sps, pps;  // avcC parameter sets (NAL payloads)
    std::vector<Mp4Sample> samples;              // decode order
    // edit list (movie time-scale for duration, media time-scale for media_time)
    struct Edit {
        uint64_t segment_duration;
        int64_t media_time;
    };
    std::vector<Edit> edits;
    uint32_t movie_timescale = 0;
};

// Parses `data` (the whole file).  Throws evc::Error.
Mp4Track mp4_parse(const std::vector<uint8_t>& data);

}  // namespace evc
