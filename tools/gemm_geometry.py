"""Tile-count x rounds x LDS / VGPR table for every GEMM shape of the ViT-B B = 32 training step, for every main-loop geometry that
exists in gemm.hip / gemm8.h and the ones the round-3 review proposed (CPU only; no GPU, no library).  The point: decide from the
geometry -- before building anything -- whether a tile form can FILL 256 CUs at these shapes.

  python tools/gemm_geometry.py > profiles/r04_gemm_geometry.txt

Columns per (shape, form): tiles = output tiles (x in-block / cross-block K splits where the form splits), slots = resident blocks
the form's LDS / VGPR budget admits on 256 CUs, rounds = tiles / slots, fill = tiles / (ceil(rounds) * slots) (share of the slot-time of
the launch that holds a tile), L1 MB = bytes the tiles stream L1 -> LDS over the whole launch, floor us = the bytes of the busiest
CU's ceil(tiles / 256) tiles through its own L1 -> LDS path at the 49 B/clk the 128-family sustains (TD busy 86 % at 42 B/clk,
DESIGN.md 4b), mfma us = FLOP at 1.95 PFLOP/s (LDS-fed MFMA clock)."""
import math

CU = 256
CLK = 1.9e9
L1_BPC = 49.0


class Form:
    def __init__(self, name, bm, bn, waves, blocks_per_cu, lds_kib, vgpr, ksplit=1, note=""):
        self.name, self.bm, self.bn, self.waves, self.bpc, self.lds, self.vgpr, self.ksplit, self.note = name, bm, bn, waves, blocks_per_cu, lds_kib, vgpr, ksplit, note


FORMS = [
    Form("128x128 4w x3/CU (persistent, today)", 128, 128, 4, 3, 48, 168),
    Form("64x128 4w x3/CU (today, small grids)", 64, 128, 4, 3, 40, 168),
    Form("64x128 8w in-block K/2 x2/CU (VAR 3, today)", 64, 128, 8, 2, 48, 128, ksplit=1, note="two 4-wave groups share one tile"),
    Form("256x128 4w x2/CU (MI 8, today)", 256, 128, 4, 2, 64, 256),
    Form("256x256 8w x1/CU (gemm8, today)", 256, 256, 8, 1, 160, 249),
    Form("128x128 8w in-block K/2 x2/CU (review item 1 i)", 128, 128, 8, 2, 80, 128, note="needs <= 128 VGPRs: acc 64 + ONE k-substep of fragments"),
    Form("256x128 8w counted-vmcnt x1/CU (gemm8 halved)", 256, 128, 8, 1, 128, 160),
    Form("256x128 8w x2/CU (one LDS stage, half-step fragments)", 256, 128, 8, 2, 80, 128),
]

M_ENC, M_DEC, M_HEAD = 5120, 50176, 45056
SHAPES = [
    # name, M, N, K, launches per step
    ("enc qkv fwd", M_ENC, 2304, 768, 12), ("enc proj fwd / dproj", M_ENC, 768, 768, 24), ("enc fc1 fwd / dfc2+dgelu", M_ENC, 3072, 768, 24),
    ("enc fc2 fwd / dfc1", M_ENC, 768, 3072, 24), ("enc dqkv", M_ENC, 768, 2304, 12),
    ("dec qkv fwd", M_DEC, 1152, 384, 4), ("dec proj fwd / dproj", M_DEC, 384, 384, 8), ("dec fc1 fwd / dfc2+dgelu", M_DEC, 1536, 384, 8),
    ("dec fc2 fwd / dfc1", M_DEC, 384, 1536, 8), ("dec dqkv", M_DEC, 384, 1152, 4),
    ("head fwd", M_HEAD, 1536, 384, 1), ("head dgrad", M_HEAD, 384, 1536, 1), ("patch embed", M_ENC, 768, 1536, 1), ("enc->dec", M_ENC, 384, 768, 1),
]
# weight gradients (TN): output P x Q, reduction R tokens; one encoder block = 4 problems, grouped g blocks per launch
WG_ENC = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
WG_DEC = [(1152, 384), (384, 384), (1536, 384), (384, 1536)]


def row(form, tiles, flop, k):
    slots = form.bpc * CU
    rounds = tiles / slots
    fill = tiles / (math.ceil(rounds) * slots)
    l1 = tiles * (form.bm + form.bn) * k * 2
    floor_us = math.ceil(tiles / CU) * ((form.bm + form.bn) * k * 2) / (L1_BPC * CLK) * 1e6      # the busiest CU's tiles through its own L1
    return f"{tiles:6d} {slots:5d} {rounds:6.2f} {fill:5.2f} {l1 / 1e6:8.0f} {floor_us:8.1f} {flop / 1.95e15 * 1e6:8.1f}"


def main():
    print(__doc__)
    hdr = f"{'form':56s} {'LDS':>4s} {'VGPR':>4s} | {'tiles':>6s} {'slots':>5s} {'rounds':>6s} {'fill':>5s} {'L1 MB':>8s} {'floor us':>8s} {'mfma us':>8s}"
    for name, M, N, K, n in SHAPES:
        print(f"\n== {name}: M {M} N {N} K {K} ({n} launches / step, {2.0 * M * N * K / 1e9:.1f} GFLOP each)")
        print(hdr)
        for f in FORMS:
            tiles = math.ceil(M / f.bm) * math.ceil(N / f.bn)
            print(f"{f.name:56s} {f.lds:4d} {f.vgpr:4d} | " + row(f, tiles, 2.0 * M * N * K, K) + (f"   [{f.note}]" if f.note else ""))
    for tag, probs, R, groups in (("encoder", WG_ENC, M_ENC, (1, 2, 3, 4, 5, 6, 7, 12)), ("decoder", WG_DEC, M_DEC, (1,))):
        for g in groups:
            flop = sum(2.0 * p * q * R for p, q in probs) * g
            print(f"\n== weight gradients, {tag}: {g} block(s) per grouped launch, reduction over {R} token rows ({flop / 1e9:.1f} GFLOP)")
            print(hdr)
            for f in FORMS:
                if "in-block" in f.name or "64x128" in f.name:
                    continue
                base = sum(math.ceil(p / f.bm) * math.ceil(q / f.bn) for p, q in probs) * g
                # token-reduction splits (f32 atomics) only where the runtime splits today: decoder groups, >= 4096 rows per split
                splits = 1 if tag == "encoder" else max(1, min(math.ceil(f.bpc * CU * 0.98 / base), R // 4096))
                pad = sum(math.ceil(p / f.bm) * f.bm * math.ceil(q / f.bn) * f.bn for p, q in probs) / sum(p * q for p, q in probs)
                print(f"{f.name:56s} {f.lds:4d} {f.vgpr:4d} | " + row(f, base * splits, flop, R / splits) + f"   splits {splits}, padded MFMA work x{pad:.2f}")


if __name__ == "__main__":
    main()
