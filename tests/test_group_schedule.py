"""The exchange of the in-library multi-GPU group (csrc/rtiow_group.hip) as a SCHEDULE, checked without GPUs.

rtiow_group_gather asks HIP and RCCL for everything through one table of calls (GatherBackend); the debug hook
rtiow_debug_gather_schedule runs the same schedule function against a recorder.  No N > 1 transport has run on
hardware yet (a one-GPU box cannot), so what CAN be pinned here is pinned: which call, on which device, stream and
communicator, with which counts and offsets, in which order, fenced by which events -- for the RCCL send/recv form
and for the peer-copy form, including ranks without rows and failing calls.

New work: the reference is single-GPU (/root/reference/src/GlobalFloatCUDAInOneWeekend/main.cu:81).
"""
import numpy as np
import pytest

SET_DEVICE, WAIT, RECORD, COPY, COPY_PEER, GROUP_START, GROUP_END, SEND, RECV = range(1, 10)
STREAM_SYNC, COPY_VIA_HOST, CARRIED_BY, DRAIN, AWAIT, ABORT_COMMS = 10, 11, 12, 13, 14, 15
G0 = 99


def stream(k):
    return 1 + k


def done(k):
    return 100 + k


def comm(k):
    return 200 + k


def rows_of(native, H, n, strip):
    return [len(native.shard_rows(H, r, n, strip)) for r in range(n)]


def offsets_bytes(rows, W, es):
    off, out = 0, []
    for r in rows:
        out.append(off * es)
        off += r * W * 3
    return out


def device_at(rec, index):
    """The device in effect when call `index` was made (field 1 of every record, as tracked by the recorder)."""
    return rec[index][1]


@pytest.mark.parametrize("prec", (32, 64))
@pytest.mark.parametrize("n,H,strip", ((8, 1080, 2), (4, 1080, 2), (2, 1080, 8), (3, 100, 7), (5, 11, 8)))
def test_rccl_schedule_counts_offsets_comms_and_fences(native, n, H, strip, prec):
    W, es = 1920, prec // 8
    devices = list(range(n))
    rows = rows_of(native, H, n, strip)
    assert sum(rows) == H
    rec, rc = native.debug_gather_schedule(devices, rows, W, prec, native.GATHER_RCCL)
    assert rc == 0
    ops = [r[0] for r in rec]
    # ---- opening fence on device 0: stream 0 waits for every other rank's render, THEN records g0
    i_g0 = next(i for i, r in enumerate(rec) if r[0] == RECORD and r[2] == G0)
    assert rec[0][:3] == (SET_DEVICE, -1, 0)
    waits0 = [r for r in rec[:i_g0] if r[0] == WAIT]
    assert [(r[2], r[3]) for r in waits0] == [(stream(0), done(k)) for k in range(1, n)]
    assert rec[i_g0][3] == stream(0) and device_at(rec, i_g0) == 0
    # ---- every other rank's stream waits for g0 on ITS device before the group opens
    i_start, i_end = ops.index(GROUP_START), ops.index(GROUP_END)
    assert ops.count(GROUP_START) == ops.count(GROUP_END) == 1 and i_g0 < i_start < i_end
    pre = rec[i_g0 + 1:i_start]
    assert [r[0] for r in pre] == [SET_DEVICE, WAIT] * (n - 1)
    for k in range(1, n):
        sd, w = pre[2 * (k - 1)], pre[2 * (k - 1) + 1]
        assert sd[2] == devices[k] and (w[2], w[3]) == (stream(k), G0) and w[1] == devices[k]
    # ---- inside the group: one send + one recv per rank WITH rows, in rank order; nothing else
    inside = rec[i_start + 1:i_end]
    live = [k for k in range(n) if rows[k] > 0]
    assert [r[0] for r in inside] == [SEND, RECV] * len(live)
    off = offsets_bytes(rows, W, es)
    for j, k in enumerate(live):
        s, r = inside[2 * j], inside[2 * j + 1]
        count = rows[k] * W * 3
        # send: rank k's framebuffer, its own communicator and stream, to peer 0
        assert s[2:] == (k, count, int(prec == 64), 0, comm(k), stream(k))
        # recv: rank 0's communicator and stream, from peer k, into rank k's block of the staging buffer
        assert r[2:] == (off[k], count, int(prec == 64), k, comm(0), stream(0))
    # blocks tile the staging buffer exactly (rank-major, no gaps, no overlap)
    ends = [off[k] + rows[k] * W * 3 * es for k in live]
    assert [off[k] for k in live][1:] == ends[:-1] and ends[-1] == W * H * 3 * es
    # ---- completion (sends / receives fail or stall AFTER they were enqueued): every sender's stream is awaited on its device
    # with its communicator's asynchronous error state, stream 0 (the receives) last
    tail = rec[i_end + 1:-1]
    awaited = [k for k in range(n - 1, -1, -1) if rows[k] > 0 or k == 0]
    assert [r[0] for r in tail] == [SET_DEVICE, AWAIT] * len(awaited)
    for j, k in enumerate(awaited):
        sd, aw = tail[2 * j], tail[2 * j + 1]
        assert sd[2] == devices[k] and aw[1] == devices[k] and (aw[2], aw[3]) == (stream(k), comm(k))
    # ---- the schedule ends with device 0 current (the caller launches the de-interleave there)
    assert rec[-1][:3] == (SET_DEVICE, rec[-2][1] if len(rec) > 1 else -1, 0)
    assert COPY not in ops and COPY_PEER not in ops


@pytest.mark.parametrize("prec", (32, 64))
def test_peer_schedule_event_choreography(native, prec):
    """Distinct devices: rank k's copy runs on ITS device and stream, after g0; done[k] is re-recorded behind the copy
    and stream 0 waits for it.  Ranks that share device 0 (and rank 0) copy on stream 0."""
    W, H, strip, es = 640, 90, 4, prec // 8
    devices = [0, 1, 2, 0, 3]                       # rank 3 shares device 0
    n = len(devices)
    rows = rows_of(native, H, n, strip)
    rec, rc = native.debug_gather_schedule(devices, rows, W, prec, native.GATHER_PEER)
    assert rc == 0
    ops = [r[0] for r in rec]
    assert GROUP_START not in ops and SEND not in ops and RECV not in ops
    i_g0 = next(i for i, r in enumerate(rec) if r[0] == RECORD and r[2] == G0)
    assert [(r[2], r[3]) for r in rec[:i_g0] if r[0] == WAIT] == [(stream(0), done(k)) for k in range(1, n)]
    off = offsets_bytes(rows, W, es)
    i_aw = ops.index(AWAIT) - 1                     # the completion phase starts with the set-device in front of the first await
    body = rec[i_g0 + 1:i_aw]
    pos = 0
    for k in range(n):
        nbytes = rows[k] * W * 3 * es
        if k == 0 or devices[k] == devices[0]:
            sd, cp = body[pos], body[pos + 1]
            assert sd[:3] == (SET_DEVICE, sd[1], 0)
            assert cp[0] == COPY and cp[1] == 0 and cp[2:6] == (off[k], k, nbytes, stream(0))
            pos += 2
        else:
            sd, w, cp, ev, sd0, w0 = body[pos:pos + 6]
            assert sd[0] == SET_DEVICE and sd[2] == devices[k]
            assert w[0] == WAIT and (w[2], w[3]) == (stream(k), G0) and w[1] == devices[k]          # not before the timed region opens
            assert cp[0] == COPY_PEER and cp[1] == devices[k] and cp[2:] == (off[k], k, nbytes, stream(k), 0, devices[k])
            assert ev[0] == RECORD and (ev[2], ev[3]) == (done(k), stream(k)) and ev[1] == devices[k]   # behind the copy, same stream
            assert sd0[0] == SET_DEVICE and sd0[2] == 0
            assert w0[0] == WAIT and (w0[2], w0[3]) == (stream(0), done(k)) and w0[1] == 0              # stream 0 sees the block
            pos += 6
    assert pos == len(body) and rec[-1][0] == SET_DEVICE and rec[-1][2] == 0
    # completion: every stream that carries a copy ACROSS devices is awaited on its device (no communicator), then stream 0
    tail = rec[i_aw:-1]
    across = [k for k in range(n - 1, 0, -1) if devices[k] != devices[0] and rows[k] > 0] + [0]
    assert [r[0] for r in tail] == [SET_DEVICE, AWAIT] * len(across)
    for j, k in enumerate(across):
        assert tail[2 * j][2] == devices[k] and tail[2 * j + 1][1:4] == (devices[k], stream(k), 0)


def test_ranks_without_rows_take_no_part(native):
    """More ranks than strips: the idle ranks neither send nor are received from (RCCL), nor copy (peer)."""
    W, H, strip, n = 64, 10, 4, 6                   # 3 strips: ranks 3..5 have no rows
    rows = rows_of(native, H, n, strip)
    assert rows == [4, 4, 2, 0, 0, 0]
    rec, rc = native.debug_gather_schedule(list(range(n)), rows, W, 32, native.GATHER_RCCL)
    assert rc == 0
    assert sorted(r[2] for r in rec if r[0] == SEND) == [0, 1, 2] and sorted(r[5] for r in rec if r[0] == RECV) == [0, 1, 2]
    rec, rc = native.debug_gather_schedule(list(range(n)), rows, W, 32, native.GATHER_PEER)
    assert rc == 0
    assert [r[3] for r in rec if r[0] in (COPY, COPY_PEER)] == [0, 1, 2]
    # the opening fence still covers every rank's render: an idle rank's done event is recorded by rtiow_group_render all the same
    i_g0 = next(i for i, r in enumerate(rec) if r[0] == RECORD and r[2] == G0)
    assert len([r for r in rec[:i_g0] if r[0] == WAIT]) == n - 1


def test_single_rank_group_sends_to_itself(native):
    """N = 1 is what a one-GPU box executes on hardware (tests/test_group.py): rank 0 sends to itself inside the group."""
    rec, rc = native.debug_gather_schedule([0], [48], 100, 32, native.GATHER_RCCL)
    assert rc == 0
    assert [r[0] for r in rec] == [SET_DEVICE, RECORD, GROUP_START, SEND, RECV, GROUP_END, SET_DEVICE, AWAIT, SET_DEVICE]
    assert rec[3][2:] == (0, 48 * 100 * 3, 0, 0, comm(0), stream(0)) and rec[4][2:] == (0, 48 * 100 * 3, 0, 0, comm(0), stream(0))


def test_a_failing_call_stops_the_schedule_and_closes_the_group(native):
    devices, rows, W = [0, 1, 2, 3], [8, 8, 8, 8], 32
    full, _ = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_RCCL)
    ops_full = [r[0] for r in full]
    for fail_at in range(len(full)):
        rec, rc = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_RCCL, fail_at)
        ops = [r[0] for r in rec]
        assert rc == 999
        if ops_full[fail_at] in (GROUP_START, SEND, RECV):
            # inside an open RCCL group a failure still closes the group (ncclGroupEnd); the communicators are then aborted (a communicator
            # that has failed is in no defined state) and nothing else follows
            assert ops == ops_full[:fail_at + 1] + [GROUP_END, ABORT_COMMS]
        else:
            assert ops == ops_full[:fail_at + 1] + [ABORT_COMMS]
    full, _ = native.debug_gather_schedule([0, 1, 0], [8, 8, 8], W, 32, native.GATHER_PEER)
    for fail_at in range(len(full)):
        rec, rc = native.debug_gather_schedule([0, 1, 0], [8, 8, 8], W, 32, native.GATHER_PEER, fail_at)
        assert rc == 999 and [r[0] for r in rec] == [r[0] for r in full][:fail_at + 1]        # no communicators in play: nothing to abort


def test_host_staged_schedule(native):
    """RTIOW_GATHER_HOST, the last resort: per rank with rows -- its device current, its stream awaited, one blocking copy through the host
    into its block of the staging buffer on device 0's stream; the blocks tile the buffer."""
    devices, rows, W = [0, 1, 2, 1], [8, 6, 0, 2], 10
    rec, rc = native.debug_gather_schedule(devices, rows, W, 64, native.GATHER_HOST)
    assert rc == 0
    copies = [r for r in rec if r[0] == COPY_VIA_HOST]
    assert [c[3] for c in copies] == [0, 1, 3]                                   # ranks with rows, in order
    off = 0
    for c, k in zip(copies, (0, 1, 3)):
        assert c[2] == off * 8 and c[4] == rows[k] * W * 3 * 8 and c[5] == stream(0) and c[6] == devices[0] and c[7] == devices[k]
        i = rec.index(c)
        assert rec[i - 1][0] == STREAM_SYNC and rec[i - 1][2] == stream(k) and rec[i - 2][0] == SET_DEVICE and rec[i - 2][2] == devices[k]
        off += rows[k] * W * 3
    assert rec[-1][0] == SET_DEVICE and rec[-1][2] == devices[0]


def test_fallback_chain_at_gather_time(native):
    """VERDICT r03 #6a: the first ncclGroupEnd / send / recv between distinct devices has never run; if it fails the frame must still arrive.
    With RTIOW_GATHER_AUTO's chain a failing transport is followed -- in the same call -- by a drain of every device and the NEXT transport
    from the top of its schedule: RCCL -> peer copies -> host-staged copies.  Pinned here against the recording call table, for a failure at
    every call of every transport."""
    devices, rows, W = [0, 1, 2, 3], [8, 8, 8, 8], 32
    rccl, _ = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_RCCL)
    peer, _ = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_PEER)
    host, _ = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_HOST)
    ops = lambda recs: [r[0] for r in recs]
    # nothing fails: the chain is the plain schedule + the closing record
    rec, rc = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_RCCL, fallback=True)
    assert rc == 0 and ops(rec[:-1]) == ops(rccl) and rec[-1][0] == CARRIED_BY and rec[-1][2] == native.GATHER_RCCL
    # RCCL fails at its k-th call: the group is closed if it was open, the devices drained, peer copies run in full
    for fail_at in range(len(rccl)):
        rec, rc = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_RCCL, fail_at, fallback=True)
        o = ops(rec)
        i = o.index(DRAIN)
        assert rc == 0 and o.count(DRAIN) == 1 and rec[i][2] == len(devices)
        head = ops(rccl)[:fail_at + 1] + ([GROUP_END] if ops(rccl)[fail_at] in (GROUP_START, SEND, RECV) else []) + [ABORT_COMMS]   # aborted BEFORE the drain
        assert o[:i] == head and o[i + 1:-1] == ops(peer)
        assert [r[2:] for r in rec[i + 1:-1]] == [r[2:] for r in peer]             # same arguments as a plain peer gather
        assert rec[-1][0] == CARRIED_BY and rec[-1][2] == native.GATHER_PEER and rec[-1][3] > 0   # and a note that says why
    # peer copies fail at their k-th call: host-staged copies carry the image
    for fail_at in range(len(peer)):
        rec, rc = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_PEER, fail_at, fallback=True)
        o = ops(rec)
        i = o.index(DRAIN)
        assert rc == 0 and o[:i] == ops(peer)[:fail_at + 1] and o[i + 1:-1] == ops(host) and rec[-1][2] == native.GATHER_HOST
    # the host path is the end of the chain: its failure is the call's failure
    rec, rc = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_HOST, 4, fallback=True)
    assert rc == 999 and DRAIN not in ops(rec) and rec[-1][2] == native.GATHER_HOST
    # without the chain (a transport requested outright) the first failure is the result (test_a_failing_call_stops_the_schedule_...)
    rec, rc = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_RCCL, 3)
    assert rc == 999 and DRAIN not in ops(rec)


def test_enqueue_ok_completion_fails(native):
    """ncclSend / ncclRecv / ncclGroupEnd and hipMemcpyPeerAsync mostly fail or stall asynchronously -- every call returns
    success and the error (or nothing at all) comes later.  The exchange's COMPLETION is therefore part of the schedule: each stream is awaited
    (stream query + ncclCommGetAsyncError + a deadline in the real table), and a failure there takes the same chain as a failing call --
    RCCL's communicators aborted before the devices are drained, then the next transport from the top."""
    devices, rows, W = [0, 1, 2, 3], [8, 8, 8, 8], 32
    ops = lambda recs: [r[0] for r in recs]
    rccl, _ = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_RCCL)
    peer, _ = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_PEER)
    host, _ = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_HOST)
    awaits = [i for i, r in enumerate(rccl) if r[0] == AWAIT]
    assert len(awaits) == 4 and all(i > ops(rccl).index(GROUP_END) for i in awaits)          # every call of the group has returned success by then
    for fail_at in awaits:
        rec, rc = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_RCCL, fail_at, fallback=True)
        o = ops(rec)
        i = o.index(DRAIN)
        assert rc == 0 and o[:i] == ops(rccl)[:fail_at + 1] + [ABORT_COMMS] and o[i + 1:-1] == ops(peer) and rec[-1][2] == native.GATHER_PEER
        # requested outright: the failure is the call's, the communicators are aborted all the same
        rec, rc = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_RCCL, fail_at)
        assert rc == 999 and ops(rec) == ops(rccl)[:fail_at + 1] + [ABORT_COMMS]
    pawaits = [i for i, r in enumerate(peer) if r[0] == AWAIT]
    assert len(pawaits) == 4
    for fail_at in pawaits:
        rec, rc = native.debug_gather_schedule(devices, rows, W, 32, native.GATHER_PEER, fail_at, fallback=True)
        o = ops(rec)
        i = o.index(DRAIN)
        assert rc == 0 and ABORT_COMMS not in o and o[:i] == ops(peer)[:fail_at + 1] and o[i + 1:-1] == ops(host) and rec[-1][2] == native.GATHER_HOST
    # both asynchronous failures in one frame: RCCL's completion fails, then the peer copies' completion fails: the host path carries the image
    assert AWAIT not in ops(host)


def test_schedule_hook_rejects_bad_arguments(native):
    with pytest.raises(native.RtiowError):
        native.debug_gather_schedule([0, 1], [4, 4], 0, 32, native.GATHER_RCCL)
    with pytest.raises(native.RtiowError):
        native.debug_gather_schedule([0, 1], [4, 4], 8, 16, native.GATHER_RCCL)
    with pytest.raises(native.RtiowError):
        native.debug_gather_schedule([0, 1], [4, 4], 8, 32, native.GATHER_AUTO)
