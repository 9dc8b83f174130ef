"""streams.stream_beside: the stream it returns runs beside the reference stream (measured the same way the helper measures),
and BagsInFlight's streams run beside each other."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stream_beside_returns_a_concurrent_stream():
    from multimodalfusion_amd import streams
    dev = torch.device("cuda", 0)
    cur = torch.cuda.current_stream(dev)
    got = [streams.stream_beside([cur], dev) for _ in range(3)]
    for st in got:
        assert st.cuda_stream != cur.cuda_stream
        assert streams._runs_beside(cur, st, dev)
    # a stream never runs beside itself: the probe must say so (its kernel queues behind the reference work)
    assert not streams._runs_beside(got[0], got[0], dev)


def test_bags_in_flight_streams_are_mutually_concurrent():
    from multimodalfusion_amd import streams
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
    from multimodalfusion_amd.pipeline import BagsInFlight
    dev = torch.device("cuda", 0)
    model = MIL_Attention_fc_surv_path(n_classes=4).to(dev)
    pipe = BagsInFlight(model, 3, dev)
    hs = [s.cuda_stream for s in pipe.streams]
    assert len(set(hs)) == 3
    for i in range(3):
        for j in range(i):
            assert streams._runs_beside(pipe.streams[j], pipe.streams[i], dev)
