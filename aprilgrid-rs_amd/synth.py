"""Deterministic synthetic AprilGrid frames with ground truth (bench / test workload generator).

Build-owned counterpart of the reference's print-chart script (scripts/generate_aprilgrid.py:
layout facts :1062-1167): a rows x cols grid of T36H11 tags (ids row-major from the BOTTOM-left,
black 2-bit border, bits MSB-first row-major from the tag's top-left, '1' = white), black
squares of `spacing` x tag size at every tag corner, rendered under a random homography with
3x3 supersampling, optical blur, sensor noise.  Everything random derives from
splitmix64(0xA9121D ^ frame_index), so frame i is the same on every rank and every run.

Runs on any torch device (the bench renders straight into HBM).  Not part of the detection
path.
"""
import json
import math
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_MASK64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK64
    return z ^ (z >> 31)


class _Rng:
    """Tiny host-side stream of uniform doubles from splitmix64."""

    def __init__(self, seed):
        self.s = seed & _MASK64

    def u(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _MASK64
        return (splitmix64(self.s) >> 11) / float(1 << 53)

    def uniform(self, a, b):
        return a + (b - a) * self.u()


# family -> (code table in tag_families_data.inc, edge bits, border bits); src/detector.rs:369-405
FAMILY_LAYOUT = {"T16H5": ("kT16H5", 4, 2), "T25H7": ("kT25H7", 5, 2), "T25H9": ("kT25H9", 5, 2),
                 "T36H11": ("kT36H11", 6, 2), "T36H11B1": ("kT36H11", 6, 1)}


def family_codes(family="T36H11"):
    """Code words of a tag family: read from the table compiled into the product (tag_families_data.inc)."""
    import re
    txt = open(os.path.join(_HERE, "csrc", "tag_families_data.inc")).read()
    m = re.search(FAMILY_LAYOUT[family][0] + r"\[\d+\] = \{(.*?)\};", txt, re.S)
    return [int(t, 16) for t in re.findall(r"0x([0-9A-Fa-f]+)ULL", m.group(1))]


def t36h11_codes():
    return family_codes("T36H11")


def _homography(src, dst):
    """3x3 H with dst ~ H src (4 point pairs), float64."""
    A, b = [], []
    for (x, y), (u, v) in zip(src, dst):
        A.append([x, y, 1, 0, 0, 0, -u * x, -u * y]); b.append(u)
        A.append([0, 0, 0, x, y, 1, -v * x, -v * y]); b.append(v)
    h = np.linalg.solve(np.asarray(A, np.float64), np.asarray(b, np.float64))
    return np.append(h, 1.0).reshape(3, 3)


class BoardSpec:
    def __init__(self, rows=6, cols=6, spacing=0.3, first_id=0):
        self.rows, self.cols, self.spacing, self.first_id = rows, cols, spacing, first_id
        self.period = 1.0 + spacing
        self.width = cols * self.period + spacing
        self.height = rows * self.period + spacing

    def tag_id(self, ix, iy_from_top):
        return self.first_id + (self.rows - 1 - iy_from_top) * self.cols + ix

    def tag_corners(self, ix, iy_from_top):
        """Board coordinates (x right, y down) of the tag's 4 outer corners, TL, TR, BR, BL."""
        x0 = self.spacing + ix * self.period
        y0 = self.spacing + iy_from_top * self.period
        return [(x0, y0), (x0 + 1.0, y0), (x0 + 1.0, y0 + 1.0), (x0, y0 + 1.0)]


def random_pose(rng, spec, width, height):
    """Image positions of the board's 4 outer corners: rotation, scale, perspective jitter."""
    size = rng.uniform(0.55, 0.9) * min(width, height)
    ang = rng.uniform(0.0, 2.0 * math.pi)
    aspect = spec.width / spec.height
    hw, hh = 0.5 * size * aspect, 0.5 * size
    base = [(-hw, -hh), (hw, -hh), (hw, hh), (-hw, hh)]
    jit = 0.10 * size
    pts = []
    for (x, y) in base:
        x += rng.uniform(-jit, jit)
        y += rng.uniform(-jit, jit)
        pts.append((x * math.cos(ang) - y * math.sin(ang), x * math.sin(ang) + y * math.cos(ang)))
    xs, ys = [p[0] for p in pts], [p[1] for p in pts]
    # translate so that the board stays inside the frame with a small margin
    m = 12.0
    lo_x, hi_x = m - min(xs), width - m - max(xs)
    lo_y, hi_y = m - min(ys), height - m - max(ys)
    cx = rng.uniform(lo_x, hi_x) if hi_x > lo_x else 0.5 * (lo_x + hi_x)
    cy = rng.uniform(lo_y, hi_y) if hi_y > lo_y else 0.5 * (lo_y + hi_y)
    return [(x + cx, y + cy) for (x, y) in pts]


def _hash_noise(idx, seed):
    """Per-pixel ~N(0,1) from integer hashing (identical on CPU and GPU): sum of 8 bytes."""
    x = idx * (-7046029254386353131) + seed  # 0x9E3779B97F4A7C15 as int64, wrapping
    x = x ^ ((x >> 30) & ((1 << 34) - 1))
    x = x * (-4658895280553007687)           # 0xBF58476D1CE4E5B9
    x = x ^ ((x >> 27) & ((1 << 37) - 1))
    x = x * (-7723592293110705685)           # 0x94D049BB133111EB
    x = x ^ ((x >> 31) & ((1 << 33) - 1))
    s = torch.zeros_like(x)
    for k in range(8):
        s = s + ((x >> (8 * k)) & 0xFF)
    # sum of 8 U{0..255}: mean 1020, variance 8*(256^2-1)/12
    return (s.to(torch.float32) - 1020.0) / math.sqrt(8.0 * (256.0 ** 2 - 1.0) / 12.0)


def _gauss_blur(img, sigma):
    r = max(1, int(math.ceil(3 * sigma)))
    xs = torch.arange(-r, r + 1, device=img.device, dtype=torch.float32)
    k = torch.exp(-(xs * xs) / (2 * sigma * sigma))
    k = k / k.sum()
    x = img[None, None]
    x = torch.nn.functional.pad(x, (r, r, 0, 0), mode="replicate")
    x = torch.nn.functional.conv2d(x, k.view(1, 1, 1, -1))
    x = torch.nn.functional.pad(x, (0, 0, r, r), mode="replicate")
    x = torch.nn.functional.conv2d(x, k.view(1, 1, -1, 1))
    return x[0, 0]


def render_frame(frame_index, width, height, device="cpu", spec=None, codes=None, pure_noise=False,
                 supersample=3, family="T36H11"):
    """-> (uint8 tensor [H,W] on `device`, ground truth {tag_id: 4x2 float64 corner array TL,TR,BR,BL}).
    family: the tags' code table and cell layout (edge x edge code bits inside a black border)."""
    spec = spec or BoardSpec()
    codes = codes or family_codes(family)
    _, edge, border = FAMILY_LAYOUT[family]
    cells, nbits = edge + 2 * border, edge * edge
    seed = splitmix64(0xA9121D ^ frame_index)
    rng = _Rng(seed)
    black = rng.uniform(15.0, 50.0)
    white = rng.uniform(185.0, 235.0)
    bg = rng.uniform(88.0, 168.0)
    dev = torch.device(device)
    S = supersample
    seed_i64 = seed - (1 << 64) if seed >= (1 << 63) else seed
    idx = torch.arange(width * height, device=dev, dtype=torch.int64).view(height, width)
    gt = {}
    if pure_noise:
        img = torch.full((height, width), bg, device=dev, dtype=torch.float32) + 40.0 * _hash_noise(idx, seed_i64 ^ 0x5bd1)
    else:
        corners_img = random_pose(rng, spec, width, height)
        corners_board = [(0.0, 0.0), (spec.width, 0.0), (spec.width, spec.height), (0.0, spec.height)]
        H = _homography(corners_board, corners_img)
        Hinv = np.linalg.inv(H)
        for iy in range(spec.rows):
            for ix in range(spec.cols):
                pts = []
                for (x, y) in spec.tag_corners(ix, iy):
                    p = H @ np.array([x, y, 1.0])
                    pts.append((p[0] / p[2], p[1] / p[2]))
                gt[spec.tag_id(ix, iy)] = np.asarray(pts, np.float64)
        # bit lookup table [rows*cols tags][edge*edge]: 1.0 = white
        bits = torch.zeros((spec.rows * spec.cols, nbits), dtype=torch.float32)
        for iy in range(spec.rows):
            for ix in range(spec.cols):
                code = codes[spec.tag_id(ix, iy)]
                for c in range(nbits):
                    bits[iy * spec.cols + ix, c] = float((code >> (nbits - 1 - c)) & 1)
        bits = bits.to(dev).reshape(-1)
        hi = torch.tensor(Hinv, dtype=torch.float64, device=dev)
        off = (torch.arange(S, device=dev, dtype=torch.float64) + 0.5) / S - 0.5
        u = torch.arange(width, device=dev, dtype=torch.float64)
        v = torch.arange(height, device=dev, dtype=torch.float64)
        acc = torch.zeros((height, width), device=dev, dtype=torch.float32)
        margin = 0.5
        for dy in off.tolist():
            for dx in off.tolist():
                uu = (u + dx)[None, :]
                vv = (v + dy)[:, None]
                den = hi[2, 0] * uu + hi[2, 1] * vv + hi[2, 2]
                X = ((hi[0, 0] * uu + hi[0, 1] * vv + hi[0, 2]) / den).to(torch.float32)
                Y = ((hi[1, 0] * uu + hi[1, 1] * vv + hi[1, 2]) / den).to(torch.float32)
                on_page = (X > -margin) & (X < spec.width + margin) & (Y > -margin) & (Y < spec.height + margin)
                in_board = (X >= 0) & (X < spec.width) & (Y >= 0) & (Y < spec.height)
                ixf = torch.floor(X / spec.period)
                iyf = torch.floor(Y / spec.period)
                fx = X - ixf * spec.period
                fy = Y - iyf * spec.period
                small = in_board & (fx < spec.spacing) & (fy < spec.spacing)
                in_tag = in_board & (fx >= spec.spacing) & (fy >= spec.spacing) & (ixf < spec.cols) & (iyf < spec.rows)
                cx = torch.clamp(torch.floor((fx - spec.spacing) * float(cells)), 0, cells - 1).to(torch.int64)
                cy = torch.clamp(torch.floor((fy - spec.spacing) * float(cells)), 0, cells - 1).to(torch.int64)
                inner = (cx >= border) & (cx < border + edge) & (cy >= border) & (cy < border + edge)
                tag_lin = (torch.clamp(iyf, 0, spec.rows - 1).to(torch.int64) * spec.cols +
                           torch.clamp(ixf, 0, spec.cols - 1).to(torch.int64))
                bit_idx = tag_lin * nbits + torch.clamp(cy - border, 0, edge - 1) * edge + torch.clamp(cx - border, 0, edge - 1)
                bitv = bits[bit_idx]
                tag_val = torch.where(inner & (bitv > 0.5), torch.tensor(white, device=dev), torch.tensor(black, device=dev))
                val = torch.full_like(X, white)
                val = torch.where(small, torch.tensor(black, device=dev), val)
                val = torch.where(in_tag, tag_val, val)
                val = torch.where(on_page, val, torch.tensor(bg, device=dev))
                acc += val
        img = acc / float(S * S)
        tex = 14.0 * _hash_noise(idx // 4, seed_i64 ^ 0x77aa)  # coarse background texture
        img = img + torch.where(img == bg, tex, torch.zeros_like(tex))
    img = _gauss_blur(img, 0.8)
    img = img + 2.0 * _hash_noise(idx, seed_i64)
    out = torch.clamp(torch.round(img), 0, 255).to(torch.uint8)
    return out, gt


def render_batch(first_index, n, width, height, device="cpu", fmt="L8", pure_noise=False):
    """-> (tensor [n,H,W] uint8 | [n,H,W] int16 holding u16 bits | [n,H,W,3] uint8, list of gt)."""
    spec, codes = BoardSpec(), t36h11_codes()
    frames, gts = [], []
    for i in range(n):
        f, gt = render_frame(first_index + i, width, height, device, spec, codes, pure_noise)
        frames.append(f)
        gts.append(gt)
    x = torch.stack(frames)
    if fmt == "L8":
        return x, gts
    idx = torch.arange(x.numel(), device=x.device, dtype=torch.int64).view(x.shape)
    if fmt == "L16":
        lo = (_hash_noise(idx, 0x1234567) * 60.0).to(torch.int64)
        v = torch.clamp(x.to(torch.int64) * 257 + lo, 0, 65535)
        return (v - (v >= 32768).to(torch.int64) * 65536).to(torch.int16), gts  # u16 bit pattern
    if fmt == "RGB8":
        chans = []
        for c in range(3):
            d = torch.round(_hash_noise(idx, 0xABC0 + c) * 1.5).to(torch.int64)
            chans.append(torch.clamp(x.to(torch.int64) + d, 0, 255).to(torch.uint8))
        return torch.stack(chans, dim=-1).contiguous(), gts
    raise ValueError(fmt)
