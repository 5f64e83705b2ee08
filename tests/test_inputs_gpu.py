"""GPU: batched input generation on the device (zkwg_generate_inputs_device) vs the host mirror
zkwg.inputs (itself a restatement of packages/helpers/src/input-generators.ts:190-252)."""
import pytest

pytestmark = pytest.mark.gpu


def _host_record(c, d, N, M, selector):
    from zkwg import inputs
    inp = inputs.generate_email_verifier_inputs_from_dkim_result(d, N, M, sha_precompute_selector=selector)
    return c.pack(inp)


def test_device_input_generation_matches_host_mirror():
    import zkwg
    from zkwg import synth
    N, M = 1024, 1536
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    ds = [synth.synthetic_dkim_result(12, i, body_len=50 + 97 * i) for i in range(12)]
    recs, st = zkwg.generate_inputs_device(c, ds)
    assert st == [0] * 12
    host = recs.cpu().numpy()
    for i, d in enumerate(ds):
        assert host[i].tobytes() == _host_record(c, d, N, M, None)
    # the generated records feed the witness path directly
    wit, status = c.calculate_batch_host(host.tobytes(), want_witness=False)
    assert status == [0] * 12


def test_device_input_generation_with_selector_and_errors():
    import zkwg
    from zkwg import synth
    N, M = 1024, 512
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    # a common marker line so that one selector works for the whole batch
    ds = []
    for i in range(6):
        d = synth.synthetic_dkim_result(13, i, body_len=900)
        body = bytearray(d["body"])
        body[700:708] = b"ZKMARKER"
        d = dict(d, body=bytes(body))
        ds.append(d)
    recs, st = zkwg.generate_inputs_device(c, ds, selector="ZKMARKER")
    assert st == [0] * 6
    host = recs.cpu().numpy()
    for i, d in enumerate(ds):
        assert host[i].tobytes() == _host_record(c, d, N, M, "ZKMARKER")
    # errors: selector missing (3), remaining body too long (2), header too long (1)
    _, st = zkwg.generate_inputs_device(c, ds[:2], selector="\x01nope")
    assert st == [3, 3]
    _, st = zkwg.generate_inputs_device(c, ds[:2])
    assert st == [2, 2]
    big = dict(ds[0], headers=ds[0]["headers"] + b"x" * 600)
    _, st = zkwg.generate_inputs_device(c, [big], selector="ZKMARKER")
    assert st == [1]
    # the reference's findIndexInUint8Array is not a general substring search ("aab" in "aaab")
    from zkwg import inputs
    assert inputs.find_index_in_uint8array(b"xaaab", b"aab") == -1 and b"xaaab".find(b"aab") == 2
    d = dict(ds[0], body=b"x" * 640 + b"aaab" + b"y" * 100)
    _, st = zkwg.generate_inputs_device(c, [d], selector="aab")
    assert st == [3]
    # getAdjustedSelector (input-generators.ts:44-105, 224-227): the selector spans a "=\r\n" soft line break of the
    # quoted-printable body -- found in the cleaned content, mapped back, searched for as the body's own bytes
    soft = []
    for i, at in enumerate((700, 703, 707, 640)):
        d = synth.synthetic_dkim_result(14, i, body_len=900)
        body = bytearray(d["body"])
        body[at - 3:at + 8] = (b"ZKMARKER")[:i + 2] + b"=\r\n" + (b"ZKMARKER")[i + 2:]
        soft.append(dict(d, body=bytes(body)))
    recs, st = zkwg.generate_inputs_device(c, soft, selector="ZKMARKER")
    assert st == [0] * 4
    host = recs.cpu().numpy()
    for i, d in enumerate(soft):
        assert host[i].tobytes() == _host_record(c, d, N, M, "ZKMARKER"), i


def test_device_input_generation_remove_soft_line_breaks():
    # removeSoftLineBreaks = 1: decodedEmailBodyIn = removeSoftLineBreaks(bodyRemaining) built on the device
    import zkwg
    from zkwg import synth, inputs
    N, M = 576, 640
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0, remove_soft_line_breaks=1)
    ds = [synth.synthetic_dkim_result(14, i, body_len=200 + 60 * i, soft_breaks=True) for i in range(6)]
    assert sum(d["body"].count(b"=\r\n") for d in ds) >= 6
    recs, st = zkwg.generate_inputs_device(c, ds)
    assert st == [0] * 6
    host = recs.cpu().numpy()
    for i, d in enumerate(ds):
        inp = inputs.generate_email_verifier_inputs_from_dkim_result(d, N, M, remove_soft_line_breaks_flag=True)
        assert host[i].tobytes() == c.pack(inp)
    wit, status = c.calculate_batch_host(host.tobytes(), want_witness=False)
    assert status == [0] * 6
