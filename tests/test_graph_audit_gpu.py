"""Every captured training step is walked node by node (xpt_graph_node_census through the C ABI): a memset node -- which
replays wrongly on this runtime, DESIGN.md section 6 -- makes the trainer REFUSE the capture instead of training on
garbage.  (The reference's graph mode is @tf.function, model/train_val.py:95-102.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def hip_memset_async(tensor):
    """hipMemsetAsync on torch's current stream: what PyTorch's multi-block reductions and some library solvers do to
    their semaphores / workspaces (tools/replay_probe_memset.py) -- captured, it becomes a memset NODE."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    assert hip.hipMemsetAsync(tensor.data_ptr(), 0, tensor.numel() * tensor.element_size(),
                              torch.cuda.current_stream().cuda_stream) == 0


def test_census_counts_kernel_and_memset_nodes(gpu_device):
    from xpt_mde_2021_amd.hip import ops
    x = torch.ones(1 << 16, device=gpu_device)
    y = torch.empty_like(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        y.copy_(x * 2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    clean = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(clean):
        y.copy_(x * 2 + 1)
    c = ops.graph_census(clean)
    assert c["kernel"] >= 1 and c["memset"] == 0 and c["total"] == sum(c[k] for k in ("kernel", "memcpy", "memset", "host", "other"))
    dirty = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(dirty):
        hip_memset_async(y)                     # the node class the trainers refuse
        y.add_(x)
    d = ops.graph_census(dirty)
    assert d["memset"] >= 1 and d["kernel"] >= 1, d


def test_step_graph_refuses_a_capture_with_a_memset_node(gpu_device):
    from xpt_mde_2021_amd.model import train_val as tv
    buf = torch.ones(4096, device=gpu_device)

    def dirty_step(feats):
        hip_memset_async(buf)
        return (feats["image5d"].sum() + buf.sum(),)

    def clean_step(feats):
        return (feats["image5d"].sum() * 2,)

    feats = {"image5d": torch.rand(2, 5, 8, 8, 3, device=gpu_device)}
    with pytest.raises(RuntimeError, match="memset node"):
        tv._StepGraph(dirty_step)(feats)
    graph = tv._StepGraph(clean_step)
    out = graph(feats)
    assert graph.census["memset"] == 0 and graph.census["kernel"] >= 1
    assert torch.allclose(out[0], feats["image5d"].sum() * 2)
