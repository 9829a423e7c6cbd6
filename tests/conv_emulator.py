"""NumPy emulation of the *index algebra* of bpb_conv_igemm_kernel / bpb_conv_wgrad_kernel (csrc/conv_igemm.hip).

It re-executes, tile by tile, exactly what a workgroup does with a ConvProb / WgradProb descriptor -- halo staging
into an LDS image, per-pixel offsets, tap offsets, packed-weight addressing, output mapping, accumulate flag --
but with plain dot products instead of MFMA lanes.  This lets the CPU test-suite validate the host-side geometry
(graph.Net.conv_problem, the dgrad parity classes, tile factorisation, weight packing) without a GPU; the MFMA
fragment layout itself is validated on the GPU by tests/test_gpu_*.py.
"""
import numpy as np


def pack_fwd(w, cin_pad):
    """wf[t][ci/4][co][4] = W[co][ci][t]   (bpb_pack_weights_kernel, forward layout)."""
    cout, cin, r, s = w.shape
    t = r * s
    out = np.zeros((t, cin_pad // 4, cout, 4), dtype=w.dtype)
    wt = w.reshape(cout, cin, t)
    for ci in range(cin):
        out[:, ci // 4, :, ci % 4] = wt[:, ci, :].T
    return out.reshape(-1)


def pack_dgrad(w, cin_pad):
    """wd[t][co/4][ci][4] = W[co][ci][t]   (dgrad layout: K dim = co, N dim = ci)."""
    cout, cin, r, s = w.shape
    t = r * s
    out = np.zeros((t, cout // 4, cin_pad, 4), dtype=w.dtype)
    wt = w.reshape(cout, cin, t)
    for co in range(cout):
        out[:, co // 4, :cin, co % 4] = wt[co, :, :].T
    return out.reshape(-1)


def run_conv(p, x, wpk, y, bias=None):
    """p: object with the ConvProb fields; x [N,Hi,Wi,Cin], wpk flat packed weights, y [N,Ho,Wo,Cout] (in/out)."""
    ti_n, th_n, tw_n = 1 << p.lTI, 1 << p.lTH, 1 << p.lTW
    mt = ti_n * th_n * tw_n
    assert mt == (4 >> p.lwn) * p.mt_r * 32 and p.n_ntiles == -(-p.Cout // ((32 * p.nt) << p.lwn))
    cin, cout, ld = p.Cin, p.Cout, p.LD
    cin4 = cin // 4

    stats = np.zeros((p.n_mtiles, 2, cout))
    tiles_n = -(-p.N // ti_n)
    assert p.n_mtiles == tiles_n * p.tiles_a * p.tiles_b
    for mtile in range(p.n_mtiles):
        tb = mtile % p.tiles_b
        t2 = mtile // p.tiles_b
        ta, tn = t2 % p.tiles_a, t2 // p.tiles_a
        n0, a0, b0 = tn << p.lTI, ta << p.lTH, tb << p.lTW
        acc = np.zeros((mt, cout))
        m = np.arange(mt)
        tw = m & (tw_n - 1)
        th = (m >> p.lTW) & (th_n - 1)
        ti = m >> (p.lTW + p.lTH)
        pix = (ti * p.HH + th * p.sa) * p.HW + tw * p.sa             # halo pixel index of each output pixel
        for cb in range(0, cin, p.CK):
            halo = np.zeros((ti_n * p.HH * p.HW, ld))
            for hp in range(ti_n * p.HH * p.HW):
                t = hp // p.HW
                hc = hp - t * p.HW
                tii = t // p.HH
                hr = t - tii * p.HH
                n, ih, iw = n0 + tii, a0 * p.sa + hr + p.ih0, b0 * p.sa + hc + p.iw0
                if n < p.N and 0 <= ih < p.Hi and 0 <= iw < p.Wi:
                    halo[hp, :p.CK] = x[n, ih, iw, cb:cb + p.CK]
            for i in range(p.Rt):
                for j in range(p.St):
                    dh, dw = p.dh0 + p.dhs * i, p.dw0 + p.dws * j
                    assert dh >= 0 and dw >= 0
                    widx = p.w0 + p.wrs * i + p.wss * j
                    hidx = pix + dh * p.HW + dw
                    assert hidx.max() < halo.shape[0], 'tap reads outside the staged halo'
                    a = halo[hidx, :p.CK]                               # [pixels, CK]
                    for c in range(p.CK):
                        ci = cb + c
                        q, e = ci // 4, ci % 4
                        base = ((widx * cin4 + q) * cout) * 4
                        brow = wpk[base + np.arange(cout) * 4 + e]
                        acc += np.outer(a[:, c], brow)
        for mm in range(mt):
            n, a, b = n0 + ti[mm], a0 + th[mm], b0 + tw[mm]
            if n < p.N and a < p.A and b < p.B:
                v = acc[mm].copy()
                if bias is not None:
                    v += bias
                oh, ow = a * p.osh + p.ooh, b * p.osw + p.oow
                if p.accumulate:
                    v += y[n, oh, ow]
                y[n, oh, ow] = v
                stats[mtile, 0] += v
                stats[mtile, 1] += v * v
    return stats


def run_wgrad(p, x, dy):
    """Returns dW[t][ci][co] summed over splits exactly as the slabs + reduce would (geometry check only)."""
    ti_n, th_n, tw_n = 1 << p.lTI, 1 << p.lTH, 1 << p.lTW
    assert ti_n * th_n * tw_n == 128
    out = np.zeros((p.T, p.Cin, p.Cout))

    for mtile in range(p.n_mtiles):
        tb = mtile % p.tiles_b
        t2 = mtile // p.tiles_b
        ta, tn = t2 % p.tiles_a, t2 // p.tiles_a
        n0, a0, b0 = tn << p.lTI, ta << p.lTH, tb << p.lTW
        halo = np.zeros((ti_n * p.HH * p.HW, p.Cin))
        for hp in range(ti_n * p.HH * p.HW):
            t = hp // p.HW
            hc = hp - t * p.HW
            tii = t // p.HH
            hr = t - tii * p.HH
            n, ih, iw = n0 + tii, a0 * p.sa + hr + p.ih0, b0 * p.sa + hc + p.iw0
            if n < p.N and 0 <= ih < p.Hi and 0 <= iw < p.Wi:
                halo[hp] = x[n, ih, iw]
        m = np.arange(128)
        tw = m & (tw_n - 1)
        th = (m >> p.lTW) & (th_n - 1)
        ti = m >> (p.lTW + p.lTH)
        g = np.zeros((128, p.Cout))
        for mm in range(128):
            n, a, b = n0 + ti[mm], a0 + th[mm], b0 + tw[mm]
            if n < p.N and a < p.A and b < p.B:
                g[mm] = dy[n, a, b]
        pix = (ti * p.HH + th * p.sa) * p.HW + tw * p.sa
        for t in range(p.T):
            dh, dw = t // p.S, t % p.S
            hidx = pix + dh * p.HW + dw
            assert hidx.max() < halo.shape[0]
            out[t] += halo[hidx].T @ g
    return out
