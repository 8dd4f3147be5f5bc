"""float64 numpy model of the FACTORISED score network + hand-written VJP.  TEST INFRASTRUCTURE.

This is the algorithm the HIP kernels (two-for-one-diffusion_amd/csrc/dff_kernels.hip)
implement, restated on the CPU so every stage of the device code has a checkable twin.  It is
NOT the parity oracle (that is oracle/reference_twin.py, the reference's own materialised
formulation); tests check this model against the twin, and the kernels against both.

Algebra (SURVEY.md section 0.3 and 8a "Derived math", from models/graph_transformer.py:96,
125-129,225,235,241-257 -- edge embedding feeds edges_to_kv with no nonlinearity between):

    e_ij  = W_c (x_j - x_i) + b_c,    W_c = W_ekv W_edge (512x3),  b_c = W_ekv b_edge + b_ekv
    logit_ihj = scale (q_ih.k_jh + u_ih.x_j)  + const_j          u_ih = W_c,h^T q_ih  (3-vector)
    o_ih  = sum_j a_ihj v_jh + W_c,h (xbar_ih - x_i) + b_c,h     xbar_ih = sum_j a_ihj x_j

Folded once at load time (fold_weights): u is a linear map of the LayerNorm output
(W_u = blockdiag_h(W_c,h^T) W_q, 24 x H), and the W_c term of o goes straight through the
output projection (W_oc[:, h] = W_o[:, h-slice] W_c,h, H x 24; b_o' = b_o + W_o b_c), so W_c
itself never appears on the device:

    attn_out_i = sum_h W_o,h (sum_j a_ihj v_jh) + sum_h W_oc,h (xbar_ih - x_i) + b_o'

x enters the energy ONLY through u.x_j (logits) and xrel_ih = xbar_ih - x_i (values); node
inputs are x-independent (graph_transformer.py:100-103).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
from scipy.special import erf

HEADS, DH, INNER = 8, 64, 512
SCALE = DH ** -0.5


def fold_weights(p: Dict[str, np.ndarray], n_layers: int) -> dict:
    """Natural state-dict arrays -> the folded float64 matrices the kernel uses."""
    P = {k: np.asarray(v, np.float64) for k, v in p.items()}
    W_edge, b_edge = P["edge_embedding.weight"], P["edge_embedding.bias"]
    out = dict(W_node=P["node_embedding.weight"], b_node=P["node_embedding.bias"],
               w_dec=P["node_decoder.weight"][0], b_dec=P["node_decoder.bias"][0], layers=[])
    for l in range(n_layers):
        pre = f"graphtransformer.layers.{l}."
        a = pre + "0.0.fn."
        W_c = P[a + "edges_to_kv.weight"] @ W_edge  # (512,3)
        b_c = P[a + "edges_to_kv.weight"] @ b_edge + P[a + "edges_to_kv.bias"]  # (512,)
        Wq, bq = P[a + "to_q.weight"], P[a + "to_q.bias"]
        Wkv, bkv = P[a + "to_kv.weight"], P[a + "to_kv.bias"]
        Wo, bo = P[a + "to_out.weight"], P[a + "to_out.bias"]
        H = Wq.shape[1]
        Wu = np.zeros((HEADS * 3, H))
        bu = np.zeros(HEADS * 3)
        Woc = np.zeros((H, HEADS * 3))
        for h in range(HEADS):
            s = slice(h * DH, (h + 1) * DH)
            Wu[3 * h:3 * h + 3] = W_c[s].T @ Wq[s]
            bu[3 * h:3 * h + 3] = W_c[s].T @ bq[s]
            Woc[:, 3 * h:3 * h + 3] = Wo[:, s] @ W_c[s]
        out["layers"].append(dict(
            ln1_g=P[pre + "0.0.norm.weight"], ln1_b=P[pre + "0.0.norm.bias"],
            Wq=Wq, bq=bq, Wk=Wkv[:INNER], bk=bkv[:INNER], Wv=Wkv[INNER:], bv=bkv[INNER:],
            Wu=Wu, bu=bu, Wo=Wo, Woc=Woc, bo=bo + Wo @ b_c,
            g1=P[pre + "0.1.proj.0.weight"][0],
            ln2_g=P[pre + "1.0.norm.weight"], ln2_b=P[pre + "1.0.norm.bias"],
            W1=P[pre + "1.0.fn.0.weight"], b1=P[pre + "1.0.fn.0.bias"],
            W2=P[pre + "1.0.fn.2.weight"], b2=P[pre + "1.0.fn.2.bias"],
            g2=P[pre + "1.1.proj.0.weight"][0]))
    return out


def fold_kv(fw: dict) -> dict:
    """H == head dimension only: the same network with k = v = LayerNorm output (what dff_host.hip does when
    hidden == 64): q'_h = W_k,h^T (W_q,h n + b_q,h) (the j-constant part of the logits drops out of the softmax),
    W_o,h' = W_o,h W_v,h, b_o' += W_o b_v.  W_u / W_oc stay those of the ORIGINAL q and W_o.  forward() / backward()
    on the result give the same energies and forces; the stashed q, k, v are q', n, n."""
    out = dict(fw)
    out["layers"] = []
    for lw in fw["layers"]:
        H = lw["Wq"].shape[1]
        if H != DH:
            out["layers"].append(lw)
            continue
        n = dict(lw)
        Wq, bq, Wo = np.zeros_like(lw["Wq"]), np.zeros_like(lw["bq"]), np.zeros_like(lw["Wo"])
        eye = np.zeros_like(lw["Wk"])
        for h in range(HEADS):
            s = slice(h * DH, (h + 1) * DH)
            Wq[s] = lw["Wk"][s].T @ lw["Wq"][s]
            bq[s] = lw["Wk"][s].T @ lw["bq"][s]
            Wo[:, s] = lw["Wo"][:, s] @ lw["Wv"][s]
            eye[s] = np.eye(DH)
        n.update(Wq=Wq, bq=bq, Wk=eye, bk=np.zeros_like(lw["bk"]), Wv=eye.copy(), bv=np.zeros_like(lw["bv"]),
                 Wo=Wo, bo=lw["bo"] + lw["Wo"] @ lw["bv"])
        out["layers"].append(n)
    return out


# ----------------------------------------------------------------------------- pieces
def layer_norm(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mu) * rstd
    return xhat * g + b, xhat, rstd


def layer_norm_bwd(dy, xhat, rstd, g):
    dyg = dy * g
    return rstd * (dyg - dyg.mean(-1, keepdims=True) - xhat * (dyg * xhat).mean(-1, keepdims=True))


def gate(w, x, res):
    H = x.shape[-1]
    z = x @ w[:H] + res @ w[H:2 * H] + (x - res) @ w[2 * H:]
    g = 1.0 / (1.0 + np.exp(-z))
    return x * g[..., None] + res * (1 - g[..., None]), g


def gate_bwd(w, x, res, g, dout):
    H = x.shape[-1]
    dg = (dout * (x - res)).sum(-1)
    dz = dg * g * (1 - g)
    dx = dout * g[..., None] + dz[..., None] * (w[:H] + w[2 * H:])
    dres = dout * (1 - g[..., None]) + dz[..., None] * (w[H:2 * H] - w[2 * H:])
    return dx, dres


def gelu(x):
    return 0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))


def gelu_grad(x):
    return 0.5 * (1.0 + erf(x / math.sqrt(2.0))) + x * np.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def _heads(t):  # (B,N,512) -> (B,8,N,64)
    B, N, _ = t.shape
    return t.reshape(B, N, HEADS, DH).transpose(0, 2, 1, 3)


def _unheads(t):  # (B,8,N,64) -> (B,N,512)
    B, _, N, _ = t.shape
    return t.transpose(0, 2, 1, 3).reshape(B, N, INNER)


# ----------------------------------------------------------------------------- forward
def forward(fw: dict, x: np.ndarray, t: np.ndarray):
    """x (B,N,3) ALREADY centred, t (B,) normalised time.  Returns (energy (B,N), stash)."""
    x = np.asarray(x, np.float64)
    t = np.asarray(t, np.float64).reshape(-1)
    B, N, _ = x.shape
    Wn = fw["W_node"]  # (H, N+1): one-hot(bead) columns then the t column
    nodes = Wn[:, :N].T[None] + t[:, None, None] * Wn[:, N][None, None] + fw["b_node"]
    nodes = np.broadcast_to(nodes, (B, N, nodes.shape[-1])).copy()
    stash = []
    for lw in fw["layers"]:
        st = dict(nodes_in=nodes)
        a, st["xhat1"], st["rstd1"] = layer_norm(nodes, lw["ln1_g"], lw["ln1_b"])
        q = _heads(a @ lw["Wq"].T + lw["bq"])
        k = _heads(a @ lw["Wk"].T + lw["bk"])
        v = _heads(a @ lw["Wv"].T + lw["bv"])
        u = (a @ lw["Wu"].T + lw["bu"]).reshape(B, N, HEADS, 3).transpose(0, 2, 1, 3)  # (B,8,N,3)
        logits = SCALE * (np.einsum("bhid,bhjd->bhij", q, k) + np.einsum("bhic,bjc->bhij", u, x))
        logits -= logits.max(-1, keepdims=True)
        pr = np.exp(logits)
        pr /= pr.sum(-1, keepdims=True)
        o = np.einsum("bhij,bhjd->bhid", pr, v)
        xrel = np.einsum("bhij,bjc->bhic", pr, x) - x[:, None]  # (B,8,N,3)
        attn_out = _unheads(o) @ lw["Wo"].T + xrel.transpose(0, 2, 1, 3).reshape(B, N, 24) @ lw["Woc"].T + lw["bo"]
        nodes1, g1 = gate(lw["g1"], attn_out, nodes)
        f, st["xhat2"], st["rstd2"] = layer_norm(nodes1, lw["ln2_g"], lw["ln2_b"])
        h_pre = f @ lw["W1"].T + lw["b1"]
        ff = gelu(h_pre) @ lw["W2"].T + lw["b2"]
        nodes2, g2 = gate(lw["g2"], ff, nodes1)
        st.update(q=q, k=k, v=v, u=u, P=pr, attn_out=attn_out, nodes1=nodes1, g1=g1, h_pre=h_pre,
                  ff=ff, g2=g2, nodes2=nodes2, a=a, f=f, o=o, xrel=xrel)
        stash.append(st)
        nodes = nodes2
    energy = nodes @ fw["w_dec"] + fw["b_dec"]
    return energy, stash


# ----------------------------------------------------------------------------- backward (VJP wrt x)
def backward(fw: dict, x: np.ndarray, stash: list, intermediates: dict | None = None) -> np.ndarray:
    """d(sum energy)/dx for centred x, using only what forward() stashed."""
    x = np.asarray(x, np.float64)
    B, N, _ = x.shape
    H = fw["w_dec"].shape[0]
    dn = np.broadcast_to(fw["w_dec"], (B, N, H)).copy()  # d(sum_i e_i)/d nodes_L
    dx = np.zeros_like(x)
    for l in range(len(stash) - 1, -1, -1):
        lw, st = fw["layers"][l], stash[l]
        dff, dn1 = gate_bwd(lw["g2"], st["ff"], st["nodes1"], st["g2"], dn)
        dh = (dff @ lw["W2"]) * gelu_grad(st["h_pre"])
        df = dh @ lw["W1"]
        dn1 = dn1 + layer_norm_bwd(df, st["xhat2"], st["rstd2"], lw["ln2_g"])
        dattn, dnin = gate_bwd(lw["g1"], st["attn_out"], st["nodes_in"], st["g1"], dn1)
        G = _heads(dattn @ lw["Wo"])  # (B,8,N,64) = dE/do
        r = (dattn @ lw["Woc"]).reshape(B, N, HEADS, 3).transpose(0, 2, 1, 3)  # dE/dxrel (B,8,N,3)
        pr, u = st["P"], st["u"]
        da = np.einsum("bhid,bhjd->bhij", G, st["v"]) + np.einsum("bhic,bjc->bhij", r, x)
        ds = pr * (da - (pr * da).sum(-1, keepdims=True))
        dx += np.einsum("bhij,bhic->bjc", pr, r) + SCALE * np.einsum("bhij,bhic->bjc", ds, u)
        dx -= r.sum(1)
        if intermediates is not None:
            intermediates[f"l{l}.dn_out"] = dn
            intermediates[f"l{l}.dattn"] = dattn
            intermediates[f"l{l}.dx_acc"] = dx.copy()
        if l > 0:  # layer-0 node inputs do not depend on x: no dq/dk/dv/du needed
            dq = SCALE * np.einsum("bhij,bhjd->bhid", ds, st["k"])
            dk = SCALE * np.einsum("bhij,bhid->bhjd", ds, st["q"])
            dv = np.einsum("bhij,bhid->bhjd", pr, G)
            du = SCALE * np.einsum("bhij,bjc->bhic", ds, x)
            da_ln = (_unheads(dq) @ lw["Wq"] + _unheads(dk) @ lw["Wk"] + _unheads(dv) @ lw["Wv"]
                     + du.transpose(0, 2, 1, 3).reshape(B, N, 24) @ lw["Wu"])
            dn = dnin + layer_norm_bwd(da_ln, st["xhat1"], st["rstd1"], lw["ln1_g"])
    return dx


def score(p: Dict[str, np.ndarray], x: np.ndarray, t: np.ndarray, n_layers: int):
    """(forces (B,N,3), energy (B,N)) = the op of graph_transformer.py:77-114 in float64."""
    fw = fold_weights(p, n_layers)
    xc = np.asarray(x, np.float64)
    xc = xc - xc.mean(1, keepdims=True)
    e, st = forward(fw, xc, t)
    return -backward(fw, xc, st), e
