"""End-to-end probe of the operator entry point: host frames in, homography dict out (upload over PCIe INCLUDED).
usage: python tools/e2e_probe.py [WxH:resize_width:nframes ...]   (default 1280x720:1280:257 and 1920x1080:400:257)"""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from evenvizion_amd import synthetic as S
from evenvizion_amd.processing.video_processing import get_homography_dict
cases = sys.argv[1:] or ["1280x720:1280:257", "1920x1080:400:257"]
res = {}
for case in cases:
    wh, rw, nfr = case.split(":")
    w, h = map(int, wh.split("x")); rw = int(rw); nfr = int(nfr)
    gray, _ = S.make_stream(11, 17, w, h)                       # 17 distinct frames, walked there and back
    idx = [i % 32 if i % 32 <= 16 else 32 - i % 32 for i in range(nfr)]
    frames = [S.gray_to_bgr(gray[i][None])[0] for i in range(17)]
    seq = [frames[i] for i in idx]
    get_homography_dict(S.SyntheticCapture(seq[:66]), resize_width=rw, features_type_list=["ORB"])          # warm-up (context, first import)
    t = time.perf_counter()
    d = get_homography_dict(S.SyntheticCapture(seq), resize_width=rw, features_type_list=["ORB"])
    dt = time.perf_counter() - t
    res[case] = dict(pairs=len(d) - 1, seconds=round(dt, 3), pairs_per_s=round((len(d) - 1) / dt, 1))
print(json.dumps(res, indent=1))
