#!/usr/bin/env python3
"""Digitise EVERY result figure the reference holds into tests/golden/plot_anchors.json.

The figures are the output of the reference's src/test.cpp: one MPC::run() followed by 25 closed-loop MPC::solve()
calls on the scenario of test.cpp:45-50, i.e. 26 samples each of CTE, ePsi, delta and velocity produced by the real
IPOPT/CppAD path.  They are the only numeric IPOPT outputs that exist for this path, so they are used as
known-answer anchors at plot precision (about +-1.5 pixels):
  * 14 figures for N in {10..50} x dt in {0.1, 0.05, 0.02}      (submission-report.md:250-265)
  * 6 figures of three panels each for the cost-weight sweeps       (submission-report.md:303-327)
The fixture holds data read off output files the reference already holds; no reference source is copied.
Needs PIL and /root/reference (this container only); tests read the committed JSON.

How a panel is read: the axes frames are found as long dark horizontal runs; y ticks are the short dark marks left
of a frame, x ticks the marks below it (x = 0, 5, ..., 25); the tick LABELS cannot be read without OCR and are
listed by hand below (top to bottom, as printed in the figure); the curve is the matplotlib-blue pixels inside the
frame, sampled at the 26 integer x positions.

Caveats kept in the fixture:
  * sample 0 of the delta panel is run()'s steer value times maxSteering (test.cpp:76), which includes run()'s
    post-processing of the time the figure was made; tests skip delta[0] and rely on ePsi[0] = epsi1.
  * the weight figures are screenshots pasted side by side, labelled by hand by the author.  The labels are what
    the figure says ("label"); "note" records where the content contradicts the label (see tests/test_plot_anchors.py).
"""
import json
import os

import numpy as np
from PIL import Image

REF = "/root/reference/examples"
HERE = os.path.dirname(os.path.abspath(__file__))

STD = {"cte": [0.1, 0.0, -0.1], "epsi": [0.030, 0.025, 0.020, 0.015], "delta": [0.015, 0.010, 0.005], "v": [35, 30]}
L002 = {"cte": [-0.150, -0.175, -0.200, -0.225], "epsi": [0.02, 0.01, 0.00], "delta": [0.01, 0.00], "v": [29, 28, 27]}
L005 = {"cte": [-0.1, -0.2], "epsi": [0.02, 0.01], "delta": [0.015, 0.010, 0.005, 0.000], "v": [32, 30, 28]}
# tick labels, top to bottom, read off each figure by eye
NDT = {
    "10-01-2": (10, 0.1, STD),
    "20-01-2": (20, 0.1, {"cte": [0.0, -0.1, -0.2], "epsi": [0.08, 0.06, 0.04, 0.02], "delta": [0.02, 0.01, 0.00], "v": [35, 30]}),
    "30-01-2": (30, 0.1, {"cte": [0, -1, -2], "epsi": [0.15, 0.10, 0.05], "delta": [0.04, 0.02, 0.00], "v": [35, 30]}),
    "40-01-2": (40, 0.1, {"cte": [0, -2, -4, -6], "epsi": [0.3, 0.2, 0.1], "delta": [0.2, 0.1, 0.0], "v": [35, 30]}),
    "10-005-2": (10, 0.05, {"cte": [0.2, 0.0, -0.2], "epsi": [0.02, 0.01, 0.00], "delta": [0.01, 0.00], "v": [32, 30, 28]}),
    "20-005-2": (20, 0.05, {"cte": [0.0, -0.1, -0.2], "epsi": [0.02, 0.01], "delta": [0.01, 0.00], "v": [32, 30, 28]}),
    "30-005-2": (30, 0.05, L005), "40-005-2": (40, 0.05, L005), "50-005": (50, 0.05, L005),
    "10-002": (10, 0.02, {"cte": [-0.10, -0.15, -0.20], "epsi": [0.02, 0.00, -0.02], "delta": [0.01, 0.00], "v": [29, 28, 27]}),
    "20-002": (20, 0.02, L002), "30-002": (30, 0.02, L002), "40-002": (40, 0.02, L002), "50-002": (50, 0.02, L002),
}
# weight figures: (Config::weights index, [(label on the panel, tick labels)] left to right)
WEIGHTS = {
    "ePsi-weights": (1, [(1, STD), (100, {"cte": [0.0, -0.1], "epsi": [0.030, 0.025, 0.020], "delta": [0.015, 0.010, 0.005], "v": [35, 30]}),
                         (1000, {"cte": [0.5, 0.0], "epsi": [0.02, 0.01], "delta": [0.01, 0.00], "v": [35, 30]})]),
    "velocity-weights": (2, [(1, STD), (100, STD), (0, {"cte": [0.2, 0.0, -0.2], "epsi": [0.02, 0.00], "delta": [0.015, 0.010, 0.005], "v": [20, 10]})]),
    "delta-weights": (3, [(1, STD), (300, STD), (5000, {"cte": [0.5, 0.0], "epsi": [0.02, 0.00], "delta": [0.015, 0.010, 0.005], "v": [35, 30]})]),
    "delta-delta-weights": (4, [(0, STD), (1500, STD), (5000, {"cte": [0.2, 0.0, -0.2], "epsi": [0.030, 0.025, 0.020, 0.015], "delta": [0.015, 0.010, 0.005], "v": [35, 30]})]),
    "accel-weights": (6, [(0, STD), (100, STD), (10000, STD)]),
    "delta-accel-weights": (7, [(0, STD), (100, STD), (5000, STD)]),
}


# What reproduces each panel (established with the oracle, tests/test_plot_anchors.py asserts every line of it):
#  * every figure was made with Config::weights of config-stable.json EXCEPT WEIGHT_DDELTA = 1500 (the json in the
#    repository says 1200; the report recommends "between 1000 and 1500"): with 1500 all 13 reproducible N/dt figures and
#    all 18 weight panels agree within 2.1 pixels, with 1200 ePsi sits 3-5 pixels off throughout;
#  * three hand-written panel labels do not describe their content: in ePsi-weights.png the labels 1 and 100 are
#    interchanged, and the panels "Weight = 1" of delta-weights.png and "Weight = 0" of delta-delta-weights.png show the
#    default run (they are pixel-identical to the default panels next to them).
BASE_WEIGHT_OVERRIDE = {4: 1500.0}
REPRODUCED_WITH = {("ePsi-weights", 1): 100, ("ePsi-weights", 100): 1, ("delta-weights", 1): 300, ("delta-delta-weights", 0): 1500}
# examples/40-01-2.png (N=40, dt=0.1: a 4 s horizon, twice the fitted stretch of road): reproduced up to sample 11; at
# sample 12 the reference's curve jumps (delta 0.244, then -0.08), which is IPOPT returning an unconverged iterate at
# its 0.5 s max_cpu_time (MPC.cpp:176-178; the solves of this figure take 80-210 iterations) -- not reproducible.
REPRODUCIBLE_SAMPLES = {"40-01-2": 12}


def clusters(idx, gap=1):
    """centres of runs of consecutive indices"""
    idx = np.asarray(idx)
    if len(idx) == 0:
        return []
    cuts = np.where(np.diff(idx) > gap)[0]
    starts = np.r_[0, cuts + 1]; ends = np.r_[cuts, len(idx) - 1]
    return [0.5 * (idx[a] + idx[b]) for a, b in zip(starts, ends)]


def find_frames(dark, min_len):
    """axes frames = pairs of long horizontal dark runs with the same extent: [(top, bottom, left, right)]"""
    H, W = dark.shape
    lines = []
    for r in range(H):
        row = dark[r]
        if row.sum() < min_len:
            continue
        xs = np.where(row)[0]
        cuts = np.where(np.diff(xs) > 1)[0]
        starts = np.r_[0, cuts + 1]; ends = np.r_[cuts, len(xs) - 1]
        for a, b in zip(starts, ends):
            if xs[b] - xs[a] >= min_len:
                lines.append((r, xs[a], xs[b]))
    # runs of one row that are separated by a small gap are one border (a curve touching it splits the run); borders
    # drawn 2-3 pixels thick (the screenshots) are reduced to their first row
    merged = []
    for r, a, b in sorted(lines):
        if merged and merged[-1][0] == r and a - merged[-1][2] < 30:
            merged[-1] = (r, merged[-1][1], b)
        else:
            merged.append((r, a, b))
    lines = []
    for r, a, b in merged:
        if any(r - r0 in (1, 2, 3) and abs(a - a0) <= 3 and abs(b - b0) <= 3 for r0, a0, b0 in lines[-12:]):
            continue
        lines.append((r, a, b))
    frames = []
    used = set()
    for i, (r, x0, x1) in enumerate(lines):
        if i in used:
            continue
        for j in range(i + 1, len(lines)):
            r2, y0, y1 = lines[j]
            overlap = min(x1, y1) - max(x0, y0)
            if j not in used and overlap > 0.8 * max(x1 - x0, y1 - y0) and r2 - r > 30:
                frames.append((r, r2, max(x0, y0), min(x1, y1))); used.add(i); used.add(j)   # a tick on a border row extends it outwards
                break
    return frames


def read_panel(im, dark, blue, frame, labels):
    top, bot, left, right = frame
    # y ticks: dark marks just left of the frame; x ticks: just below it
    ycol = dark[top:bot + 1, max(0, left - 4):left - 1].all(axis=1) if left >= 4 else np.zeros(bot - top + 1, bool)
    yt = [top + c for c in clusters(np.where(ycol)[0])]
    assert len(yt) == len(labels), ("y ticks", len(yt), labels)
    xrow = dark[bot + 2:bot + 5, left:right + 1].all(axis=0)
    xt = [left + c for c in clusters(np.where(xrow)[0])]
    # the title of the panel below can reach into these rows: keep the outermost marks (x = 0 and 25) and check that the
    # other four sit where they should
    want = [xt[0] + k * (xt[-1] - xt[0]) / 5.0 for k in range(6)]
    assert all(min(abs(w - t) for t in xt) <= 1.5 for w in want), ("x ticks 0,5,..,25", xt)
    xt = want
    per_px = (labels[0] - labels[-1]) / (yt[-1] - yt[0])
    vals = []
    for i in range(26):
        c = xt[0] + i * (xt[-1] - xt[0]) / 25.0
        ci = int(round(c))
        rows = np.where(blue[top + 2:bot - 1, ci])[0]
        if len(rows) == 0:                                       # the end samples sit on the line's cap: look one pixel inside
            for dc in (1, -1, 2, -2):
                rows = np.where(blue[top + 2:bot - 1, ci + dc])[0]
                if len(rows):
                    break
        row = top + 2 + 0.5 * (rows.min() + rows.max())
        vals.append(labels[0] - (row - yt[0]) * per_px)
    return [float(v) for v in vals], float(per_px)


def read_figure(path, panel_labels):
    """panel_labels: list (left to right) of dicts name -> tick labels.  Returns one dict of curves per sub-figure."""
    im = np.array(Image.open(path).convert("RGB")).astype(int)
    dark = im.sum(axis=2) < 200
    blue = (abs(im[:, :, 0] - 31) < 45) & (abs(im[:, :, 1] - 119) < 45) & (abs(im[:, :, 2] - 180) < 45)
    frames = find_frames(dark, min_len=min(400, im.shape[1] // (3 * len(panel_labels)) ))
    # group into sub-figures by the left edge, order top to bottom inside each
    lefts = sorted(set(f[2] for f in frames))
    groups = []
    for l in lefts:
        if not groups or l - groups[-1][0] > 50:
            groups.append([l])
    cols = []
    for g in groups:
        fs = sorted([f for f in frames if abs(f[2] - g[0]) <= 50], key=lambda f: f[0])
        if len(fs) >= 5:
            cols.append(fs[:5])
    assert len(cols) == len(panel_labels), (path, len(cols), [len(c) for c in cols])
    out = []
    for fs, labels in zip(cols, panel_labels):
        curves, px = {}, {}
        for name, f in zip(("cte", "epsi", "delta", "v"), fs[:4]):
            curves[name], px[name] = read_panel(im, dark, blue, f, labels[name])
        out.append({"curves": curves, "pixel_value": px})
    return out


def main():
    res = {"source": "examples/*.png of the reference repository (output of its src/test.cpp with IPOPT/CppAD)", "samples": 26,
           "scenario": "src/test.cpp:45-50 with config-stable.json; N/dt or one weight changed as stated per entry",
           "base_weight_override": {str(k): v for k, v in BASE_WEIGHT_OVERRIDE.items()}, "n_dt": {}, "weights": {}}
    for name, (N, dt, labels) in NDT.items():
        r = read_figure(os.path.join(REF, name + ".png"), [labels])[0]
        r.update({"N": N, "dt": dt, "file": "examples/%s.png" % name, "reproducible_samples": REPRODUCIBLE_SAMPLES.get(name, 26)})
        res["n_dt"][name] = r
        print(name, "cte0 %.4f epsi0 %.4f delta1 %.4f v25 %.3f" % (r["curves"]["cte"][0], r["curves"]["epsi"][0], r["curves"]["delta"][1], r["curves"]["v"][25]))
    for name, (idx, panels) in WEIGHTS.items():
        if panels is None:
            continue
        rs = read_figure(os.path.join(REF, name + ".png"), [p[1] for p in panels])
        for (label, _), r in zip(panels, rs):
            r.update({"weight_index": idx, "label": label, "reproduced_with": REPRODUCED_WITH.get((name, label), label),
                      "file": "examples/%s.png" % name})
            print(name, label, "cte25 %.4f epsi0 %.4f delta1 %.4f v25 %.3f" % (r["curves"]["cte"][25], r["curves"]["epsi"][0], r["curves"]["delta"][1], r["curves"]["v"][25]))
        res["weights"][name] = rs
    json.dump(res, open(os.path.join(HERE, "plot_anchors.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
