"""The generated model directories are the reference's graphs (structure) in the reference's file format."""
import os

import numpy as np
import pytest

from conftest import REFERENCE
from tools import gen_models, ncnn_param

NAMED_OUTPUTS = {
    ("rife-v4.6", "flownet"): ["flow0", "flow1", "flow2", "flow3", "out0"],     # rife.cpp:3142-3145, 2653-2669
    ("rife-v4", "flownet"): ["flow0", "flow1", "flow2", "flow3", "out0"],       # same blob contract, models/rife-v4/flownet.param
    ("rife-v2.3", "flownet"): ["flow"],                                          # rife.cpp:948-950
    ("rife-v2.3", "contextnet"): ["f1", "f2", "f3", "f4"],                       # rife.cpp:1027-1039
    ("rife-v2.3", "fusionnet"): ["output"],                                      # rife.cpp:1070-1098
    ("rife-v3.1", "flownet"): ["flow"],
    ("rife-v3.1", "contextnet"): ["f1", "f2", "f3", "f4"],
    ("rife-v3.1", "fusionnet"): ["output"],
    ("rife", "flownet"): ["flow"], ("rife", "contextnet"): ["f1", "f2", "f3", "f4"], ("rife", "fusionnet"): ["output"],
    ("rife-HD", "flownet"): ["flow"], ("rife-HD", "contextnet"): ["f1", "f2", "f3", "f4"], ("rife-HD", "fusionnet"): ["output"],
}
needs_ref = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree only exists in the build container")


@needs_ref
@pytest.mark.parametrize("fam,net", list(NAMED_OUTPUTS))
def test_generated_param_is_the_reference_graph(modeldirs, fam, net):
    ref = ncnn_param.parse(os.path.join(REFERENCE, "models", fam, net + ".param"))
    gen = ncnn_param.parse(os.path.join(modeldirs[fam], net + ".param"))
    assert len(ref) == len(gen)
    assert ncnn_param.weighted_layers(ref) == ncnn_param.weighted_layers(gen)   # same .bin stream order
    # v1 family: the negative slopes of the SE bottleneck FCs are trained constants kept in the .param (i.e. weights): not structure
    v1 = fam in ("rife", "rife-HD")
    for blob in NAMED_OUTPUTS[(fam, net)]:
        assert ncnn_param.structural_hash(ref, blob, v1) == ncnn_param.structural_hash(gen, blob, v1), blob


@needs_ref
@pytest.mark.parametrize("alias", ["rife-v2", "rife-v2.4"])
def test_v2_family_graphs_identical(alias):
    """rife-v2 / v2.4 ship byte-identical graphs to v2.3: covered for free."""
    for net in ("flownet", "contextnet", "fusionnet"):
        a = open(os.path.join(REFERENCE, "models", alias, net + ".param")).read()
        b = open(os.path.join(REFERENCE, "models", "rife-v2.3", net + ".param")).read()
        assert a == b


@needs_ref
@pytest.mark.parametrize("alias", ["rife-UHD", "rife-anime"])
def test_hd_family_graphs_identical(alias):
    """rife-UHD and rife-anime ship the rife-HD graphs byte for byte."""
    for net in ("flownet", "contextnet", "fusionnet"):
        a = open(os.path.join(REFERENCE, "models", alias, net + ".param")).read()
        b = open(os.path.join(REFERENCE, "models", "rife-HD", net + ".param")).read()
        assert a == b


@needs_ref
def test_v3_family_graphs_identical():
    """rife-v3.0 ships the same three graphs as v3.1."""
    for net in ("flownet", "contextnet", "fusionnet"):
        a = open(os.path.join(REFERENCE, "models", "rife-v3.0", net + ".param")).read()
        b = open(os.path.join(REFERENCE, "models", "rife-v3.1", net + ".param")).read()
        assert a == b


@pytest.mark.parametrize("fam", ["rife-v4.6", "rife-v2.3", "rife-v4", "rife-v3.1", "rife", "rife-HD"])
def test_weight_count_identity(modeldirs, fam):
    """param[6] == oc*ic*k*k for every conv/deconv once channels are propagated."""
    for net in gen_models.FAMILIES[fam]:
        g = gen_models.FAMILIES[fam][net]()
        for l in g.weighted():
            if "w" in l["meta"]:
                oc, ic, kh, kw = l["meta"]["w"]
                key = "2=" if l["type"] == "InnerProduct" else "6="        # weight_data_size lives at id 2 for InnerProduct
                n = [int(p.split("=")[1]) for p in l["params"] if p.startswith(key)][0]
                assert n == oc * ic * kh * kw


def test_v46_parameter_total():
    g = gen_models.ifnet_v46()
    tot = sum(int(np.prod(l["meta"]["w"])) + l["meta"]["w"][0] for l in g.weighted())
    assert tot == 5302416          # SURVEY App. A "Weight totals"
    assert len(g.weighted()) == 44


def test_oracle_reads_generated_bin_to_eof(modeldirs):
    from oracle import pyoracle
    o = pyoracle.OracleRIFE(rife_v4=True, num_threads=1)
    o.load(modeldirs["rife-v4.6"])
    used, total = o.bin_bytes(0)
    assert used == total == os.path.getsize(os.path.join(modeldirs["rife-v4.6"], "flownet.bin"))


@needs_ref
def test_oracle_reads_real_contextnet_bin_to_eof(modeldirs, tmp_path):
    """The one real trained weight file that survives in the reference (rife-v2.3/contextnet.bin, 2 387 688 B)
    must be consumed exactly by the App. D layout rule."""
    import shutil
    from oracle import pyoracle
    d = tmp_path / "v23real"
    shutil.copytree(modeldirs["rife-v2.3"], d)
    shutil.copy(os.path.join(REFERENCE, "models", "rife-v2.3", "contextnet.bin"), d / "contextnet.bin")
    o = pyoracle.OracleRIFE(rife_v2=True, num_threads=1)
    o.load(str(d))
    used, total = o.bin_bytes(1)
    assert used == total == 2387688
