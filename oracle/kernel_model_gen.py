"""ORACLE (test infrastructure only) -- float64 model of the GENERAL input branches of the score net
in the factorised "extended head" formulation the HIP kernel uses, with the hand-written VJP.

Covers every combination of the reference's input flags (models/graph_transformer.py:53-58,99-102,
116-140): use_intrinsic_coords (edge features x_j - x_i), use_distances (edge feature |x_j - x_i|^2),
use_abs_coords (node features include x).  The shipped checkpoints are (1, 0, 0) and have their own,
leaner model in kernel_model.py; main_train.py's defaults are (0, 1, 1).

With W_c = W_ekv W_edge (512 x n_edge_features) split per head into W_c3,h (64 x 3, zero without
intrinsic coords) and c_h (64, zero without distances):
    u_i = W_c3,h^T q_i        s_i = c_h . q_i
    logits_ij = scale (q_i.k_j + Qx_i . Kx_j),   Qx_i = [u_i - 2 s_i x_i, s_i],  Kx_j = [x_j, |x_j|^2]
    [o_i | m1_i | m2_i] = sum_j a_ij [v_j | x_j | |x_j|^2]
    xrel_i = m1_i - x_i                 D_i = |x_i|^2 - 2 x_i.m1_i + m2_i  (= sum_j a_ij |x_i - x_j|^2)
    attn_out_i = sum_h (o W_o,h^T + xrel W_oc,h^T + D w_od,h) + b_o'
(terms constant in j are dropped from the logits: they cancel in the softmax).
"""
from typing import Dict

import numpy as np

from .kernel_model import (HEADS, DH, INNER, SCALE, layer_norm, layer_norm_bwd, gate, gate_bwd, gelu, gelu_grad,
                           _heads, _unheads)


def fold_weights(p: Dict[str, np.ndarray], n_layers: int, n_beads: int, intr: bool, dist: bool, abs_: bool) -> dict:
    P = {k: np.asarray(v, np.float64) for k, v in p.items()}
    W_edge, b_edge = P["edge_embedding.weight"], P["edge_embedding.bias"]
    nf = 3 * intr + dist
    assert W_edge.shape[1] == (nf if nf else 1)
    Wn = P["node_embedding.weight"]
    assert Wn.shape[1] == n_beads + 1 + 3 * abs_
    out = dict(W_node=Wn, b_node=P["node_embedding.bias"], w_dec=P["node_decoder.weight"][0],
               b_dec=P["node_decoder.bias"][0], layers=[], flags=(intr, dist, abs_), N=n_beads)
    for l in range(n_layers):
        pre = f"graphtransformer.layers.{l}."
        a = pre + "0.0.fn."
        Wekv = P[a + "edges_to_kv.weight"]
        W_c = Wekv @ W_edge if nf else np.zeros((INNER, 0))
        b_c = Wekv @ b_edge + P[a + "edges_to_kv.bias"]
        W_c3 = W_c[:, :3] if intr else np.zeros((INNER, 3))
        c = W_c[:, 3 * intr] if dist else np.zeros(INNER)
        Wq, bq = P[a + "to_q.weight"], P[a + "to_q.bias"]
        Wkv, bkv = P[a + "to_kv.weight"], P[a + "to_kv.bias"]
        Wo, bo = P[a + "to_out.weight"], P[a + "to_out.bias"]
        H = Wq.shape[1]
        Wu, bu = np.zeros((HEADS, 3, H)), np.zeros((HEADS, 3))
        Ws, bs = np.zeros((HEADS, H)), np.zeros(HEADS)
        Woc, wod = np.zeros((HEADS, H, 3)), np.zeros((HEADS, H))
        for h in range(HEADS):
            s = slice(h * DH, (h + 1) * DH)
            Wu[h], bu[h] = W_c3[s].T @ Wq[s], W_c3[s].T @ bq[s]
            Ws[h], bs[h] = c[s] @ Wq[s], c[s] @ bq[s]
            Woc[h], wod[h] = Wo[:, s] @ W_c3[s], Wo[:, s] @ c[s]
        out["layers"].append(dict(
            ln1_g=P[pre + "0.0.norm.weight"], ln1_b=P[pre + "0.0.norm.bias"],
            Wq=Wq, bq=bq, Wk=Wkv[:INNER], bk=bkv[:INNER], Wv=Wkv[INNER:], bv=bkv[INNER:],
            Wu=Wu, bu=bu, Ws=Ws, bs=bs, Wo=Wo, Woc=Woc, wod=wod, bo=bo + Wo @ b_c,
            g1=P[pre + "0.1.proj.0.weight"][0],
            ln2_g=P[pre + "1.0.norm.weight"], ln2_b=P[pre + "1.0.norm.bias"],
            W1=P[pre + "1.0.fn.0.weight"], b1=P[pre + "1.0.fn.0.bias"],
            W2=P[pre + "1.0.fn.2.weight"], b2=P[pre + "1.0.fn.2.bias"],
            g2=P[pre + "1.1.proj.0.weight"][0]))
    return out


def forward(fw: dict, x: np.ndarray, t: np.ndarray):
    """x (B,N,3) ALREADY centred, t (B,).  Returns (energy (B,N), stash)."""
    x = np.asarray(x, np.float64)
    t = np.asarray(t, np.float64).reshape(-1)
    B, N, _ = x.shape
    intr, dist, abs_ = fw["flags"]
    Wn = fw["W_node"]   # columns: one-hot (N) | x (3, if abs) | t
    nodes = Wn[:, :N].T[None] + t[:, None, None] * Wn[:, -1][None, None] + fw["b_node"]
    nodes = np.broadcast_to(nodes, (B, N, nodes.shape[-1])).copy()
    if abs_:
        nodes = nodes + x @ Wn[:, N:N + 3].T
    x2 = (x * x).sum(-1)                                            # |x_j|^2  (B,N)
    Kx = np.concatenate([x, x2[..., None]], -1)                     # (B,N,4): extension of K_ext and V_ext
    stash = []
    for lw in fw["layers"]:
        st = dict(nodes_in=nodes)
        a, st["xhat1"], st["rstd1"] = layer_norm(nodes, lw["ln1_g"], lw["ln1_b"])
        q = _heads(a @ lw["Wq"].T + lw["bq"])
        k = _heads(a @ lw["Wk"].T + lw["bk"])
        v = _heads(a @ lw["Wv"].T + lw["bv"])
        u = np.einsum("bnc,hkc->bhnk", a, lw["Wu"]) + lw["bu"][None, :, None, :]       # (B,8,N,3)
        s = np.einsum("bnc,hc->bhn", a, lw["Ws"]) + lw["bs"][None, :, None]            # (B,8,N)
        Qx = np.concatenate([u - 2.0 * s[..., None] * x[:, None], s[..., None]], -1)   # (B,8,N,4)
        logits = SCALE * (np.einsum("bhid,bhjd->bhij", q, k) + np.einsum("bhic,bjc->bhij", Qx, Kx))
        logits -= logits.max(-1, keepdims=True)
        pr = np.exp(logits)
        pr /= pr.sum(-1, keepdims=True)
        o = np.einsum("bhij,bhjd->bhid", pr, v)
        m = np.einsum("bhij,bjc->bhic", pr, Kx)                                        # [m1 | m2]
        m1, m2 = m[..., :3], m[..., 3]
        xrel = m1 - x[:, None]
        D = x2[:, None] - 2.0 * np.einsum("bic,bhic->bhi", x, m1) + m2
        attn_out = (_unheads(o) @ lw["Wo"].T + np.einsum("bhic,hkc->bik", xrel, lw["Woc"])
                    + np.einsum("bhi,hk->bik", D, lw["wod"]) + lw["bo"])
        nodes1, g1 = gate(lw["g1"], attn_out, nodes)
        f, st["xhat2"], st["rstd2"] = layer_norm(nodes1, lw["ln2_g"], lw["ln2_b"])
        h_pre = f @ lw["W1"].T + lw["b1"]
        ff = gelu(h_pre) @ lw["W2"].T + lw["b2"]
        nodes2, g2 = gate(lw["g2"], ff, nodes1)
        st.update(q=q, k=k, v=v, u=u, s=s, Qx=Qx, P=pr, m1=m1, attn_out=attn_out, nodes1=nodes1, g1=g1,
                  h_pre=h_pre, ff=ff, g2=g2, D=D)
        stash.append(st)
        nodes = nodes2
    return nodes @ fw["w_dec"] + fw["b_dec"], stash


def backward(fw: dict, x: np.ndarray, stash: list) -> np.ndarray:
    """d(sum energy)/dx for centred x."""
    x = np.asarray(x, np.float64)
    B, N, _ = x.shape
    intr, dist, abs_ = fw["flags"]
    H = fw["w_dec"].shape[0]
    x2 = (x * x).sum(-1)
    Kx = np.concatenate([x, x2[..., None]], -1)
    dn = np.broadcast_to(fw["w_dec"], (B, N, H)).copy()
    dx = np.zeros_like(x)
    for l in range(len(stash) - 1, -1, -1):
        lw, st = fw["layers"][l], stash[l]
        dff, dn1 = gate_bwd(lw["g2"], st["ff"], st["nodes1"], st["g2"], dn)
        dh = (dff @ lw["W2"]) * gelu_grad(st["h_pre"])
        df = dh @ lw["W1"]
        dn1 = dn1 + layer_norm_bwd(df, st["xhat2"], st["rstd2"], lw["ln2_g"])
        dattn, dnin = gate_bwd(lw["g1"], st["attn_out"], st["nodes_in"], st["g1"], dn1)
        G = _heads(dattn @ lw["Wo"])                                   # dE/do
        r = np.einsum("bik,hkc->bhic", dattn, lw["Woc"])                # dE/dxrel (B,8,N,3)
        gD = np.einsum("bik,hk->bhi", dattn, lw["wod"])                 # dE/dD    (B,8,N)
        pr, s = st["P"], st["s"]
        dx -= r.sum(1)
        dx += (gD[..., None] * (2.0 * x[:, None] - 2.0 * st["m1"])).sum(1)
        Gx = np.concatenate([r - 2.0 * gD[..., None] * x[:, None], gD[..., None]], -1)   # [dm1 | dm2]
        da = np.einsum("bhid,bhjd->bhij", G, st["v"]) + np.einsum("bhic,bjc->bhij", Gx, Kx)
        dS = SCALE * pr * (da - (pr * da).sum(-1, keepdims=True))
        dVx = np.einsum("bhij,bhic->bhjc", pr, Gx)                      # extension of dV_ext
        dKx = np.einsum("bhij,bhic->bhjc", dS, st["Qx"])                # extension of dK_ext
        dQx = np.einsum("bhij,bjc->bhic", dS, Kx)                       # extension of dQ_ext: [A | B]
        for ext in (dVx, dKx):
            dx += ext[..., :3].sum(1) + 2.0 * x * ext[..., 3].sum(1)[..., None]
        A, Bq = dQx[..., :3], dQx[..., 3]
        dx += (-2.0 * s[..., None] * A).sum(1)
        if l > 0 or abs_:
            dq = np.einsum("bhij,bhjd->bhid", dS, st["k"])
            dk = np.einsum("bhij,bhid->bhjd", dS, st["q"])
            dv = np.einsum("bhij,bhid->bhjd", pr, G)
            du = A
            ds = -2.0 * np.einsum("bic,bhic->bhi", x, A) + Bq
            da_ln = (_unheads(dq) @ lw["Wq"] + _unheads(dk) @ lw["Wk"] + _unheads(dv) @ lw["Wv"]
                     + np.einsum("bhic,hck->bik", du, lw["Wu"]) + np.einsum("bhi,hk->bik", ds, lw["Ws"]))
            dn = dnin + layer_norm_bwd(da_ln, st["xhat1"], st["rstd1"], lw["ln1_g"])
    if abs_:
        dx += dn @ fw["W_node"][:, fw["N"]:fw["N"] + 3]
    return dx


def score(p, x, t, n_layers, intr, dist, abs_):
    """(forces (B,N,3), energy (B,N)) in float64."""
    xc = np.asarray(x, np.float64)
    xc = xc - xc.mean(1, keepdims=True)
    fw = fold_weights(p, n_layers, xc.shape[1], intr, dist, abs_)
    e, st = forward(fw, xc, t)
    return -backward(fw, xc, st), e
