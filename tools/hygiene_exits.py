"""Execution counts (gcov) of the rarely taken exits of the oracle's SIFT / SURF statements that VERDICT r03 listed:
usage: hygiene_exits.py <dir with *.gcov>.  For an `if (...) return / continue` on one line gcov's -b output gives the branch
counts underneath; this prints the line count and the taken-branch counts."""
import os
import re
import sys

WANT = {
    "evz_sift.cpp.gcov": [
        ("adjust_local_extrema: solve with d == 0 (X stays 0)", r"if \(d != 0\) \{"),
        ("adjust_local_extrema: converged (|x| < 0.5) -> break", r"std::fabs\(xi\) < 0\.5f && std::fabs\(xr\) < 0\.5f"),
        ("adjust_local_extrema: offset beyond INT_MAX/3 -> false", r"INT_MAX / 3\)\)$|return false;$"),
        ("adjust_local_extrema: moved out of the layer range / border -> false", r"if \(layer < 1 \|\| layer > L"),
        ("adjust_local_extrema: 5 steps without convergence -> false", r"if \(i >= SIFT_MAX_INTERP_STEPS\) return false;"),
        ("adjust_local_extrema: contrast below threshold -> false", r"CONTRAST_THRESHOLD\) return false;"),
        ("adjust_local_extrema: edge response -> false", r"if \(det <= 0 \|\| tr \* tr \* et"),
        ("calc_descriptor: radius clamp to the image diagonal", r"radius = std::min\(radius, \(int\)std::sqrt"),
    ],
    "evz_surf.cpp.gcov": [
        ("interpolate: d == 0", r"if \(d != 0\) \{"),
        ("interpolate: ok == false (rejected)", r"const bool ok = "),
        ("describe: window larger than the image -> size = -1", r"if \(srows < grad_wav_size \|\| scols < grad_wav_size\)"),
        ("describe: nangle == 0 -> size = -1", r"if \(nangle == 0\)"),
    ],
}


def main(d):
    for fn, items in WANT.items():
        path = os.path.join(d, fn)
        if not os.path.exists(path):
            print("missing", fn)
            continue
        lines = open(path, errors="replace").read().splitlines()
        for title, pat in items:
            hit = None
            for i, l in enumerate(lines):
                m = re.match(r"\s*([0-9#=\-]+\*?):\s*(\d+):(.*)", l)
                if m and re.search(pat, m.group(3)):
                    hit = i
                    break
            if hit is None:
                print("  (pattern not found) %s" % title)
                continue
            m = re.match(r"\s*([0-9#=\-]+\*?):\s*(\d+):(.*)", lines[hit])
            br = []
            for l in lines[hit + 1: hit + 14]:
                b = re.match(r"branch\s+\d+\s+(taken (\d+)|never executed)", l)
                if b:
                    br.append(b.group(2) or "0")
                elif re.match(r"\s*[0-9#=\-]+\*?:\s*\d+:", l):
                    break
            print("  %-72s line %s executed %s; branches taken %s" % (title, m.group(2), m.group(1).strip(), ",".join(br)))


if __name__ == "__main__":
    main(sys.argv[1])
