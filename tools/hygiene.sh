#!/bin/bash
# CPU-side hygiene of the test infrastructure and of the host-side capture source (no GPU):
#   1. oracle/ and evenvizion_amd/capture/ built with AddressSanitizer + UndefinedBehaviorSanitizer, run over
#      tools/hygiene_inputs.py (parity inputs, rare-exit inputs, 40 damaged copies of the reference's mp4);
#   2. oracle/ built with --coverage, same inputs, gcov branch summary + the execution counts of the exits VERDICT r03 asked
#      about (adjust_local_extrema, interpolate_keypoint, the size = -1 deletions, the descriptor radius clamp).
# Output: profiles/r04_hygiene.txt
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
W=${TMPDIR:-/tmp}/evh_hygiene
rm -rf $W && mkdir -p $W/cov $W/asan
OUT=$R/profiles/r04_hygiene.txt
SRC="evz_orb.cpp evz_homography.cpp evz_sift.cpp evz_surf.cpp"
CAPSRC="evcap_api.cpp evc_mp4.cpp evc_h264_stream.cpp evc_h264_tables.cpp evc_h264_cabac.cpp evc_h264_slice.cpp evc_h264_recon.cpp evc_h264_decoder.cpp"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
(cd $R/oracle && g++ $SAN -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread -shared -o $W/asan/libevz_oracle.so $SRC -lm)
(cd $R/evenvizion_amd/capture && g++ $SAN -std=c++17 -fPIC -shared -o $W/asan/libevcap.so $CAPSRC)
{
  echo "# r04 hygiene -- $(date -u +%Y-%m-%dT%H:%MZ), $(g++ --version | head -1)"
  echo "## 1. ASan + UBSan (oracle/*.cpp, evenvizion_amd/capture/*.cpp), tools/hygiene_inputs.py"
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) \
    EVZ_ORACLE_SO=$W/asan/libevz_oracle.so EVCAP_SO=$W/asan/libevcap.so python3 $R/tools/hygiene_inputs.py 2>&1 | tail -25
  echo "exit code: ${PIPESTATUS[0]}  (0 and no report above = clean)"
} > $OUT
(cd $W/cov && for f in $SRC; do cp $R/oracle/$f .; done && cp $R/oracle/*.h $R/oracle/*.inc . &&
 for f in $SRC; do g++ --coverage -O0 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -pthread -c $f -o ${f%.cpp}.o; done &&
 g++ --coverage -shared -pthread -o libevz_oracle.so evz_orb.o evz_homography.o evz_sift.o evz_surf.o -lm)
(cd $W/cov && EVZ_ORACLE_SO=$W/cov/libevz_oracle.so HYGIENE_FUZZ=0 python3 $R/tools/hygiene_inputs.py > run.log 2>&1; gcov -b -c evz_sift.cpp evz_surf.cpp evz_orb.cpp evz_homography.cpp > gcov.log 2>&1 || true)
{
  echo
  echo "## 2. coverage of the oracle under the same inputs (gcov -b)"
  grep -A4 "^File 'evz_" $W/cov/gcov.log | grep -v "^--" | grep -E "File|Lines|Branches|Taken"
  echo
  echo "### execution counts of the exits named in VERDICT r03 (count: source line)"
  python3 $R/tools/hygiene_exits.py $W/cov
} >> $OUT
tail -60 $OUT
