#!/usr/bin/env python3
"""Randomised parity sweep of the conv engine (forward f32 / bf16, input gradient, weight gradient) against
torch-CPU float64 on shapes the fixed test cases do not enumerate: ragged M / Cout tails, every tile code,
forced split-K factors, strides, SAME / VALID, residual + mask + activation combinations.  Dev tool; the
tolerances are the test suite's (f32: 1e-4 * max(1, |x|))."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F

from faster_rcnn_amd import ops
from oracle import keras_ref


def ref_conv(x, w, stride, padding):
    return keras_ref.conv2d(x, w, None, stride, padding, dtype=torch.float64)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    tiles = [0, 1, 2, 3, 11, 12, 13, 21, 22, 41, 42, 43, 122, 222, 322, 522, 922]
    worst = 0.0
    fails = 0
    for it in range(n_cases):
        k = int(rs.choice([1, 1, 3, 3, 5, 7]))
        stride = int(rs.choice([1, 1, 1, 2]))
        padding = "same" if k > 1 and rs.rand() < 0.8 else "valid"
        cin = int(rs.choice([3, 3, 4, 20, 32, 64, 64, 96, 128, 256, 512]))
        cout = int(rs.choice([9, 36, 40, 64, 72, 100, 128, 192, 256, 512]))
        n = int(rs.choice([1, 1, 2, 5, 40]))
        h, w = int(rs.randint(k, 40)), int(rs.randint(k, 40))
        tile = int(rs.choice(tiles))
        act = [None, "relu", "sigmoid"][rs.randint(3)]
        x = rs.randn(n, h, w, cin).astype(np.float32)
        wt = (rs.randn(k, k, cin, cout) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
        scale = (1 + 0.1 * rs.randn(cout)).astype(np.float32)
        shift = (0.1 * rs.randn(cout)).astype(np.float32)
        want = ref_conv(x, wt, stride, padding) * torch.from_numpy(scale).double() + torch.from_numpy(shift).double()
        res = None
        if rs.rand() < 0.5:
            res = rs.randn(*want.shape).astype(np.float32)
            want = want + torch.from_numpy(res).double()
        if act == "relu":
            want = want.clamp(min=0)
        elif act == "sigmoid":
            want = torch.sigmoid(want)
        pc = ops.PackedConv(wt, scale, shift)
        desc = "it=%d n=%d h=%d w=%d cin=%d cout=%d k=%d s=%d %s act=%s res=%s tile=%d" % (it, n, h, w, cin, cout, k, stride, padding, act, res is not None, tile)
        try:
            got = ops.conv2d(torch.from_numpy(x).cuda(), pc, stride, padding, act, None if res is None else torch.from_numpy(res).cuda(), tile=tile)
        except Exception as e:
            print("RAISED", desc, e); fails += 1; continue
        err = ((got.cpu().double() - want).abs() / want.abs().clamp(min=1.0)).max().item()
        worst = max(worst, err)
        if not err <= 1e-4:
            print("FAIL fwd %.3g" % err, desc); fails += 1
        # position-major twin of the same launch (tap skipping, split-K / balanced forms over the compacted walk)
        if cin % 32 == 0 and k * k <= 32 and tile >= 11 and rs.rand() < 0.5:
            try:
                ws = ops.ConvWorkspace()
                with ops.conv_workspace(ws):
                    gp = ops.conv2d(torch.from_numpy(x).cuda().permute(1, 2, 0, 3).contiguous(), pc, stride, padding, act,
                                    None if res is None else torch.from_numpy(res).cuda().permute(1, 2, 0, 3).contiguous(),
                                    tile=int(rs.choice([tile, 61, 62])), layout=1)
                e2 = ((gp.permute(2, 0, 1, 3).cpu().double() - want).abs() / want.abs().clamp(min=1.0)).max().item()
                if not e2 <= 1e-4:
                    print("FAIL pos-major %.3g" % e2, desc); fails += 1
            except Exception as e:
                print("RAISED pos-major", desc, e); fails += 1
        # backward (stride-1 layers only have an input-gradient form)
        if it % 3 == 0 and cin % 4 == 0 and cin >= 32:
            g = rs.randn(*want.shape).astype(np.float32)
            xt = torch.from_numpy(x).double().requires_grad_(True)
            wtt = torch.from_numpy(wt).double().requires_grad_(True)
            y = ref_conv(xt, wtt, stride, padding)
            y.backward(torch.from_numpy(g).double())
            dw, db = ops.conv2d_wgrad(torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda(), k, k, stride, padding)
            e_w = ((dw.cpu().double() - wtt.grad).abs().max() / wtt.grad.abs().max().clamp(min=1.0)).item()
            e_b = ((db.cpu().double() - torch.from_numpy(g).double().sum((0, 1, 2))).abs().max() / max(1.0, float(np.abs(g.sum((0, 1, 2))).max()))).item()
            if not (e_w <= 1e-4 and e_b <= 1e-4):
                print("FAIL wgrad %.3g %.3g" % (e_w, e_b), desc); fails += 1
            if stride == 1 and (padding == "same" or k == 1):
                pd = ops.PackedDgrad(wt, None)
                gx = ops.conv2d_dgrad(torch.from_numpy(g).cuda(), pd, padding)
                if tuple(gx.shape) != tuple(x.shape):
                    print("FAIL dgrad shape", tuple(gx.shape), desc); fails += 1
                else:
                    e_x = ((gx.cpu().double() - xt.grad).abs().max() / xt.grad.abs().max().clamp(min=1.0)).item()
                    if not e_x <= 1e-4:
                        print("FAIL dgrad %.3g" % e_x, desc); fails += 1
        # bf16 forward on the same shape (cin % 64 == 0 only)
        if it % 4 == 0 and cin % 64 == 0 and k * k <= 32:
            bf = lambda a: torch.from_numpy(a).to(torch.bfloat16)
            xb, wb = bf(x), bf(wt)
            wantb = ref_conv(xb.double().numpy(), wb.double().numpy(), stride, padding) * torch.from_numpy(scale).double() + torch.from_numpy(shift).double()
            pcb = ops.PackedConvBf16(wb.float(), scale, shift)
            resb = bf(rs.randn(*wantb.shape).astype(np.float32))
            want_r = (wantb + resb.double()).clamp(min=0)
            for bt in (0, 1, 2, 3, 41, 42, 43, 44, 302, 502):
                gb = ops.conv2d_bf16(xb.cuda(), pcb, stride, padding, None, None, out_f32=True, tile=bt)
                e = ((gb.cpu().double() - wantb).abs() / wantb.abs().clamp(min=1.0)).max().item()
                if not e <= 1e-4:
                    print("FAIL bf16 %.3g tile=%d" % (e, bt), desc); fails += 1
                # residual + ReLU + bf16 store: one RNE rounding of the f32 epilogue value
                gr = ops.conv2d_bf16(xb.cuda(), pcb, stride, padding, "relu", resb.cuda(), out_f32=False, tile=bt)
                e = ((gr.cpu().double() - want_r).abs() / want_r.abs().clamp(min=1.0)).max().item()
                if not e <= 2.0 ** -8 * 1.01:
                    print("FAIL bf16 residual/relu %.3g tile=%d" % (e, bt), desc); fails += 1
    print("cases %d  failures %d  worst f32 forward error %.3g" % (n_cases, fails, worst))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
