"""ncnn `.param` text format helpers.

* `parse(path)`            -> list of layer dicts in file order
* `structural_hash(...)`   -> order-independent hash of the sub-graph that produces a named blob,
                              with `Split` layers treated as aliases.  Two param files whose named
                              outputs hash equal compute the same function of the same named inputs
                              and consume the `.bin` weight stream in the same order.

Used to prove that the graphs emitted by tools/gen_models.py are the reference's graphs
(`models/rife-v4.6/flownet.param`, `models/rife-v2.3/{flownet,contextnet,fusionnet}.param`)
without copying those files into this repository.
"""
import hashlib

WEIGHTED = ("Convolution", "Deconvolution", "PReLU", "InnerProduct")


def parse(path):
    with open(path) as f:
        lines = [l.rstrip("\n") for l in f]
    assert lines[0].strip() == "7767517", "bad ncnn magic"
    nl, nb = (int(v) for v in lines[1].split())
    layers = []
    for line in lines[2:]:
        tok = line.split()
        if not tok:
            continue
        typ, name, nin, nout = tok[0], tok[1], int(tok[2]), int(tok[3])
        bottoms = tok[4:4 + nin]
        tops = tok[4 + nin:4 + nin + nout]
        params, arrays = {}, {}
        for kv in tok[4 + nin + nout:]:
            k, v = kv.split("=")
            k = int(k)
            if k <= -23300:
                vals = v.split(",")
                arrays[-k - 23300] = [float(x) for x in vals[1:]]
            else:
                params[k] = float(v)
        layers.append(dict(type=typ, name=name, bottoms=bottoms, tops=tops, params=params, arrays=arrays))
    assert len(layers) == nl, (len(layers), nl)
    return layers


def weighted_layers(layers):
    """(type, params) of the layers that own bytes in the .bin, in stream order."""
    return [(l["type"], tuple(sorted(l["params"].items()))) for l in layers if l["type"] in WEIGHTED]


def structural_hash(layers, blob, ignore_fc_slopes=False):
    """Canonical hash of the computation producing `blob`.  `ignore_fc_slopes`: leave the fused-activation slope of
    InnerProduct layers out (in the v1 family these are trained constants stored in the .param, i.e. weights)."""
    producer = {}
    widx = 0
    for l in layers:
        l = dict(l)
        if l["type"] in WEIGHTED:
            l["widx"] = widx
            widx += 1
        for i, t in enumerate(l["tops"]):
            producer[t] = (l, i)
    memo = {}

    def h(b):
        if b in memo:
            return memo[b]
        l, oi = producer[b]
        if l["type"] == "Split":
            r = h(l["bottoms"][0])
        elif l["type"] == "Input":
            r = hashlib.sha256(("Input:" + b).encode()).hexdigest()
        else:
            parts = [l["type"], str(oi), str(l.get("widx", -1))]
            parts += ["%d=%r" % (k, v) for k, v in sorted(l["params"].items())]
            if not (ignore_fc_slopes and l["type"] == "InnerProduct"):
                parts += ["%d=[%s]" % (k, ",".join(repr(x) for x in v)) for k, v in sorted(l["arrays"].items())]
            parts += [h(x) for x in l["bottoms"]]
            r = hashlib.sha256("|".join(parts).encode()).hexdigest()
        memo[b] = r
        return r

    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(10000)
    try:
        return h(blob)
    finally:
        sys.setrecursionlimit(old)
