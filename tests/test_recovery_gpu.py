"""GPU: the HIP error path of the picture pipeline (fault injection).  The kernels that wait on the device for another
kernel's progress bound every wait and report through one sticky word; collect() answers a tripped word by re-encoding the pictures in
flight from an IDR picture on a safer schedule (enc_schedule.cpp recover()) instead of ending the stream -- what
/root/reference/src/ceracoder.c:425-438 (cb_pipeline: an element error stops the application) would otherwise turn into a dead stream.
mi355enc_debug_trip_wait() sets the word exactly as a kernel whose wait ran out does."""
import numpy as np
import pytest

from ceracoder_amd import synth

pytestmark = pytest.mark.gpu


def _run(E, w, h, clip, depth, exclusive, trips=(), force_idr_at=(), **kw):
    e = E.Encoder(w, h, gop=40, fixed_qp=30, pipeline_depth=depth, exclusive=exclusive, **kw)
    out = []
    for i, (y, uv) in enumerate(clip):
        if i in trips:
            e.debug_trip_wait(12)
        e.submit(y, uv, pts=i, force_idr=i in force_idr_at)
        if e.pending > depth:
            out.append(e.collect())
    while e.pending:
        out.append(e.collect())
    return e, out


@pytest.mark.parametrize("depth", [0, 1, 2])
def test_tripped_wait_recovers_from_an_idr_picture_and_loses_nothing(E, oracle, depth):
    w, h, n = 640, 368, 20
    clip = list(synth.s2_frames(w, h, n))
    e, out = _run(E, w, h, clip, depth, True, trips=(7,))
    st = e.stats()
    assert st.recoveries == 1 and st.last_error_word == 12 and st.safe_level == 1
    assert [o[2] for o in out] == list(range(n))                          # one access unit per picture, in order
    assert [i for i, o in enumerate(out) if o[1]] == [0, 7]               # the first picture in flight when the word was seen became an IDR picture
    # the stream is what an encoder without any device-side wait produces when asked for an IDR picture there
    ref, ref_out = _run(E, w, h, clip, 0, False, force_idr_at=(7,))
    assert [o[0] for o in out] == [o[0] for o in ref_out]
    dec = oracle.Decoder()
    for au, _, _, _ in out:
        dy, duv = dec.decode(au)
    assert np.array_equal(dy, e.fetch(E.FETCH_RECON_Y)) and np.array_equal(duv, e.fetch(E.FETCH_RECON_UV))
    assert np.array_equal(dy, ref.fetch(E.FETCH_RECON_Y))
    e.close(); ref.close()


def test_second_trip_falls_back_to_one_launch_per_wavefront_step_and_a_third_ends_the_stream(E, oracle):
    w, h, n = 640, 368, 22
    clip = list(synth.s2_frames(w, h, n))
    e, out = _run(E, w, h, clip, 2, True, trips=(5, 12))
    st = e.stats()
    assert st.recoveries == 2 and st.safe_level == 2
    assert [o[2] for o in out] == list(range(n)) and [i for i, o in enumerate(out) if o[1]] == [0, 5, 12]
    ref, ref_out = _run(E, w, h, clip, 0, False, force_idr_at=(5, 12))
    assert [o[0] for o in out] == [o[0] for o in ref_out]                 # (the per-wavefront-step kernels are the cross-check forms: same bits)
    ref.close()
    e.debug_trip_wait(3)
    y, uv = clip[0]
    e.submit(y, uv, pts=n)
    with pytest.raises(E.EncoderError):
        e.collect()
    e.close()


def test_recovery_under_rate_control_keeps_the_books(E, oracle):
    """With rate control on, the pictures in flight had been picked (planned bits booked): the recovery takes the bookings back and picks
    again; the stream stays decodable and near its setpoint."""
    w, h, fps, gop, n = 1280, 720, 30, 30, 90
    clip = list(synth.s2_frames(w, h, 12))
    e = E.Encoder(w, h, fps=fps, gop=gop, bitrate_bps=3_000_000, pipeline_depth=2, exclusive=True)
    dec, sizes = oracle.Decoder(), []
    for i in range(n):
        if i == 40:
            e.debug_trip_wait(17)
        k = i % 22
        y, uv = clip[k if k < 12 else 22 - k]
        e.submit(y, uv, pts=i)
        if e.pending > 2:
            sizes.append(e.collect())
    while e.pending:
        sizes.append(e.collect())
    assert e.stats().recoveries == 1 and [s[2] for s in sizes] == list(range(n))
    for au, _, _, _ in sizes:
        dy, duv = dec.decode(au)
    assert np.array_equal(dy, e.fetch(E.FETCH_RECON_Y))
    rate = sum(len(s[0]) for s in sizes[30:]) * 8 * fps / (n - 30)
    assert abs(rate - 3e6) / 3e6 < 0.2, rate
    e.close()
