"""Canonical, JSON-able description of a recorded product Tower (hypelcnn_amd.graph) and of its variable store.

TEST INFRASTRUCTURE shared by tests/golden/make_reference_graphs.py (which records Towers by running the REFERENCE's
unchanged plugin files through the tf_slim facade, in the build container) and tests/test_reference_wiring.py (which
records them through the product's own plugins, anywhere) -- two dumps are equal iff the two graphs are the same node
for node: kinds, sources, branch scopes / kernels / widths, normaliser flags and decay, activation, dropout keep
probabilities, residual channel maps, views (channel slices, crops) and the variable table (names, shapes, trainable,
regulariser scale, initialiser family, creation order)."""
import hashlib

import numpy as np

from hypelcnn_amd import graph as G


def _init_name(v):
    fn = v.init
    q = getattr(fn, "__qualname__", str(fn))
    return q.split(".")[0]


def dump_store(store):
    return [{"name": v.name, "shape": list(v.shape), "trainable": bool(v.trainable), "l2_scale": float(v.l2_scale),
             "init": _init_name(v)} for v in store.order]


def dump_tower(tower):
    ids = {}

    def tid(t):
        """A tensor as seen from its consumers: the owner's id plus the view."""
        own = t.owner
        if id(own) not in ids:
            ids[id(own)] = f"in:{own.name}" if own.node is None else f"n{tower.nodes.index(own.node)}"
        d = {"t": ids[id(own)], "c": t.c, "hw": list(t.hw) if t.hw else None}
        if t.root is not None:
            d["ch_off"] = t.ch_off
            if t.pixmap is not None:
                d["pixmap"] = hashlib.sha1(np.asarray(t.pixmap, np.int64).tobytes()).hexdigest()[:12]
        return d

    def act(a):
        return None if a is None else [a.kind, float(a.alpha)]

    def res(rs):
        out = []
        for src, idx in rs:
            out.append({"src": tid(src), "idx": None if idx is None else [int(i) for i in np.asarray(idx).tolist()]})
        return out

    nodes = []
    for n in tower.nodes:
        if isinstance(n, G.LinearNode):
            d = {"node": "linear", "kind": n.kind, "sources": [tid(s) for s in n.sources],
                 "branches": [{"scope": b.scope, "k": b.k, "cout": b.cout, "w": b.w.name, "bias": b.bias is not None,
                               "bn": b.bn is not None} for b in n.branches],
                 "act": act(n.act), "bn_decay": float(n.bn_decay) if n.has_bn else None,
                 "bn_eps": float(n.bn_eps) if n.has_bn else None, "training": bool(n.training) if n.has_bn else None,
                 "dropout_keep": n.dropout_keep, "residuals": res(n.residuals),
                 "in_slices": [list(s) for s in n.in_slices] if n.in_slices else None}
        elif isinstance(n, G.PostNode):
            d = {"node": "post", "src": tid(n.src), "act": act(n.act), "dropout_keep": n.dropout_keep,
                 "residuals": res(n.residuals)}
        elif isinstance(n, G.LRNNode):
            d = {"node": "lrn", "src": tid(n.src), "radius": n.radius, "bias": n.bias, "alpha": n.alpha, "beta": n.beta}
        elif isinstance(n, G.GeneratorNode):
            d = {"node": "generator", "src": tid(n.src), "only_encoder": bool(n.only_encoder),
                 "kernels": [w.shape[0] for w in n.weights], "weights": [w.name for w in n.weights]}
        elif isinstance(n, G.DenseStackNode):
            d = {"node": "dense_stack", "src": tid(n.src), "widths": n.widths, "alpha": float(n.alpha),
                 "leaky": [bool(l) for _, _, l in n.layers], "weights": [w.name for w in n.weights]}
        elif isinstance(n, G.FeatStackNode):
            d = {"node": "feat_stack", "srcs": [tid(s) for s in n.srcs]}
        else:
            raise TypeError(type(n))
        d["out"] = {"c": n.out.c, "hw": list(n.out.hw) if n.out.hw else None}
        nodes.append(d)
    return {"is_training": bool(tower.is_training), "n_dropout": tower.n_dropout, "nodes": nodes}


def dump_output(t):
    """The tensor a builder returned (a SymTensor or a FlatTensor)."""
    if isinstance(t, G.FlatTensor):
        return {"flat": [{"c": s.c, "hw": list(s.hw) if s.hw else None} for s in t.sources]}
    return {"c": t.c, "hw": list(t.hw) if t.hw else None}
