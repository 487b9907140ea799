/* evcap.h -- C ABI of libevcap.so, the capture source in front of the homography hot path ("decode").
 *
 * It replaces, for H.264 video in an MP4/MOV container, the capture object the reference opens and reads:
 *     cv2.VideoCapture(path)          /root/reference/evenvizion/examples/evenvizion_component.py:132
 *     capture.read() -> (ok, BGR)     /root/reference/evenvizion/processing/video_processing.py:58, :70
 * Host-only C++ (no GPU, no third-party code): ISO-BMFF demultiplexer + an H.264 decoder written from ITU-T Rec. H.264
 * for progressive 4:2:0 8-bit CABAC streams (the reference's own test_video.mp4 is High@3.1, CABAC, 8x8 transform,
 * B pictures, weighted prediction).  Every entry point returns a status; nothing throws across the boundary.
 *
 * All functions are thread-compatible (one handle per thread).  Sizes are in pixels, strides in bytes.
 */
#ifndef EVCAP_H
#define EVCAP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct evcap evcap;

enum {
    EVCAP_OK = 0,
    EVCAP_EOF = 1,          /* capture.read() -> (False, None): no more frames */
    EVCAP_ERR_IO = -1,      /* file cannot be opened / read */
    EVCAP_ERR_FORMAT = -2,  /* not an MP4 with an AVC track, or a stream outside the decoder's scope */
    EVCAP_ERR_STREAM = -3,  /* the bitstream could not be followed (message in evcap_last_error) */
    EVCAP_ERR_INVALID = -4  /* bad argument */
};

/* BGR conversion of the decoded 4:2:0 picture (what cv2.VideoCapture hands to Python):
 *   EVCAP_BGR_SWSCALE_X86  libswscale's unscaled yuv420p->bgr24 converter as its x86 SIMD template computes it
 *                          (ITU-R BT.601 limited range, 13-bit coefficients, chroma sample shared by a 2x2 luma quad):
 *                          what an x86-64 OpenCV 3.4 wheel runs.  Default.
 *   EVCAP_BGR_SWSCALE_C    the same converter as libswscale's portable C tables compute it (differs by at most a few
 *                          grey levels); kept to measure how sensitive results are to the conversion. */
enum { EVCAP_BGR_SWSCALE_X86 = 0, EVCAP_BGR_SWSCALE_C = 1 };

/* cv2.VideoCapture(path) -- evenvizion_component.py:132.  *out is NULL on failure; the message of the failure is kept
 * and returned by evcap_last_error(NULL). */
int evcap_open(const char* path, evcap** out);
/* Same, from a buffer holding the whole file (copied). */
int evcap_open_memory(const uint8_t* data, uint64_t size, evcap** out);
void evcap_close(evcap* c);

/* cv2.CAP_PROP_FRAME_WIDTH / HEIGHT / FRAME_COUNT / FPS -- processing_visualization.py uses them on the same object.
 * frame_count is the number of samples of the track; any pointer may be NULL. */
int evcap_info(evcap* c, int* width, int* height, int* frame_count, double* fps);

int evcap_set_bgr_mode(evcap* c, int mode);
/* Edit lists (ISO/IEC 14496-12 8.6.6): by default a sample whose composition time lies outside the track's single edit is
 * decoded but not presented -- what FFmpeg's mov demuxer under cv2.VideoCapture does, and the reason the reference's own
 * run of test_video.mp4 holds 120 pairs for 122 samples.  on = 0 presents every decoded picture.  Call before reading. */
int evcap_set_honour_edit_list(evcap* c, int on);

/* capture.read() -- video_processing.py:58,70.  Writes the next frame in presentation order as 8-bit BGR, rows of
 * width*3 bytes at `stride`.  Returns EVCAP_OK, EVCAP_EOF or an error. */
int evcap_read_bgr(evcap* c, uint8_t* dst, int64_t stride);
/* The same frame as planar 4:2:0 (Y width x height, Cb/Cr ((width+1)/2) x ((height+1)/2)); for tests. */
int evcap_read_yuv420(evcap* c, uint8_t* y, int64_t ystride, uint8_t* cb, uint8_t* cr, int64_t cstride);
/* Picture order count, decode index and slice type (0 P, 1 B, 2 I) of the frame most recently returned. */
int evcap_last_frame_info(evcap* c, int* poc, int* decode_index, int* slice_type);

/* Which coding tools the stream has exercised so far: fills up to n counters, returns how many exist.  Order:
 * macroblocks, I4x4, I8x8, I16x16, I_PCM, P_Skip, B_Skip, B_Direct_16x16, other inter, 8x8-transform macroblocks,
 * bi-predicted blocks, explicitly weighted blocks, implicitly weighted blocks, quadrants split below 8x8,
 * temporal-direct calls, spatial-direct calls, MMCO operations, list modifications, long-term pictures,
 * P slices, B slices, I slices, highest reference index used. */
int evcap_stats(evcap* c, int64_t* out, int n);

/* Message of the last failure on this handle (or of the last failed open when c is NULL).  Never NULL. */
const char* evcap_last_error(evcap* c);

#ifdef __cplusplus
}
#endif
#endif
