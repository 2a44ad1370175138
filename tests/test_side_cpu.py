"""side.py without a GPU: ``fork`` is a no-op context (the enclosed code runs where it is), ``mark`` / ``wait`` / ``join`` do nothing,
and the model-side prefetch returns without touching anything -- the CPU paths (oracle comparisons, gloo tests) never see a stream."""
import torch as th


def test_fork_is_a_no_op_without_a_gpu():
    from dualmessagepassing_amd import side
    if th.cuda.is_available():
        import pytest
        pytest.skip("the GPU behaviour is covered by tests/test_gpu_side_stream.py")
    ran = []
    with side.fork() as forked:
        assert forked is False and not side.active()
        ran.append(1)
        side.mark("a")
    side.wait("a")
    side.join()
    assert ran == [1] and not side.pending()


def test_prefetch_declines_without_gates_or_gpu():
    from dualmessagepassing_amd import dmpnn, side

    class _M:
        use_fused = True
    # no rep-net attributes, no gates: nothing to do, nothing raised, nothing pending
    assert dmpnn.prefetch_joint_indexes(_M(), None, None, None, None) is None
    assert not side.pending()
