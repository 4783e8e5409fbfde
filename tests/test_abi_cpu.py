"""CPU tests of the boundary: the C-ABI library loads without a GPU and exports every symbol
include/cactus_hip.h declares (no compute calls here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="cactus_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cactus_hip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import cactus_amd.hip as H
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(H.lib, n), f"{n} declared in include/cactus_hip.h but not exported"
        assert n in H.SIGNATURES, f"{n} has no ctypes signature in cactus_amd/hip.py"
    assert sorted(H.SIGNATURES) == names


def test_graph_library_exports_every_declared_symbol():
    import cactus_amd.hip as H
    import cactus_amd.graph as G
    names = _declared("cactus_hip_graph.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(H.lib, n), f"{n} declared in include/cactus_hip_graph.h but not exported"
    assert sorted(G.GRAPH_SIGNATURES) == names


def test_steps_library_exports_every_declared_symbol():
    import cactus_amd.hip as H
    names = _declared("cactus_hip_steps.h")
    assert names == ["cactus_hip_decode_step_dev", "cactus_hip_decode_step_split", "cactus_hip_decode_step_workspace_bytes",
                     "cactus_hip_prefill_chunk_dev", "cactus_hip_prefill_chunk_workspace_bytes", "cactus_hip_prefill_dev",
                     "cactus_hip_prefill_workspace_bytes"]
    for n in names:
        assert hasattr(H.lib, n), f"{n} declared in include/cactus_hip_steps.h but not exported"


def test_step_driver_rejects_a_bad_model_description_without_gpu():
    """Argument checks of the step driver run on the host before anything is enqueued."""
    import ctypes as C
    from cactus_amd import steps as ST
    import cactus_amd.hip as H
    w = ST._Weights()
    st = ST._DecodeState()
    assert H.lib.cactus_hip_decode_step_dev(C.byref(w), C.byref(st), None, None, 1, None) == -1
    assert b"null model description" in H.lib.cactus_hip_last_error()
    assert H.lib.cactus_hip_decode_step_workspace_bytes(C.byref(w), 0) == 0
    H.lib.cactus_hip_clear_error()


def test_graph_builder_errors_without_gpu():
    """Shape validation happens at build time on the host (graph_builder.cpp:65-76): same
    conditions, same messages, no GPU involved."""
    import pytest
    import cactus_amd.graph as G
    g = G.CactusGraph()
    a = g.input((2, 3), G.FP16)
    b = g.input((4, 2), G.FP16)
    with pytest.raises(Exception, match="Matrix dimensions incompatible"):
        g.matmul(a, b, False)
    c = g.input((2, 3, 1), G.FP16)
    with pytest.raises(Exception, match="requires 2D tensors"):
        g.matmul(c, a, False)
    i8 = g.input((2, 3), G.INT8)
    with pytest.raises(Exception, match="quantize_activations requires FP16 input"):
        g.quantize_activations(i8)
    assert g.get_node_count() == 4
    g.close()


def test_runtime_queries_work_without_gpu():
    import cactus_amd.hip as H
    assert H.lib.cactus_hip_device_count() >= 0
    assert isinstance(H.lib.cactus_hip_last_error(), bytes)


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under cactus_amd/ may import, link or call it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cactus_amd")):
        if os.path.basename(dirpath) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                if "oracle" in open(os.path.join(dirpath, f), errors="ignore").read():
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
