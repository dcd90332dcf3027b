"""The message cursor decided on the device (include/mp3s.h (iv-c): mp3s_select_plan, mp3s_rate_select_dev; csrc/k_chain.hpp
k_chain_select).  The plan is host arithmetic (CPU tests); the selection is checked against the oracle encoder -- reference
encoder/MP3_Encoder.py:808-809 (the cursor), :1154-1168 and :1257-1263 (where the message is read) -- on the cases the walk
has to get right: messages of one, two, three bits (the "bits left" variants), messages ending on every offset inside a unit,
silence inside the reach (units that take no tables), a start in silence that the plan does not cover (the chain check fails
and the host resolves it), several streams in one batch, blocks that continue a message."""
import os

import numpy as np
import pytest


# ------------------------------------------------------------------------------------------------ the plan (host only)
def test_plan_layout_and_limits(mlib):
    pat = mlib.select_patterns()
    assert pat.tolist() == [b for v in range(8) for b in ((v >> 2) & 1, (v >> 1) & 1, v & 1, 0)]
    segs = np.zeros(5, dtype=mlib.CHAIN_SEG_DTYPE)
    #          plain         not hiding      too many entries for the capacity   short stream      continuing block, 5 bits left
    segs["first_frame"] = [0, 100, 150, 9000, 9010]
    segs["n_frames"] = [100, 50, 8850, 10, 40]
    segs["hide_base"] = [32, 132, 132, 100132, 100232]
    segs["hide_begin"] = [32, 132, 132, 100132, 100232 + 995]
    segs["hide_end"] = [132, 132, 100132, 100232, 100232 + 1000]
    spans, unit, cursor = mlib.select_plan(segs, 1 << 18)            # (stream 2 would take 35 400 units x 8 to 10 entries)
    reach0 = 100 * 5 // 14 + 32

    def tail_first(bits_left, reach):                                 # MP3S_SELECT_TAIL_FIRST
        return 0 if bits_left < 2 else min((bits_left - 2) // 3, reach)

    def entries(bits_left, reach):                                    # MP3S_SELECT_ENTRIES
        return 8 * reach + 2 * (reach - tail_first(bits_left, reach))
    t0 = tail_first(100, reach0)
    assert t0 == 32 and entries(100, reach0) == 8 * reach0 + 2 * (reach0 - 32)
    assert spans["reach"].tolist() == [reach0, 0, 0, 40, 5 * 5 // 14 + 32]
    assert mlib.select_plan(segs, 1 << 20)[0]["reach"].tolist() == [reach0, 0, 35400, 40, 5 * 5 // 14 + 32]
    n0, n3, n4 = entries(100, reach0), entries(100, 40), entries(5, 33)
    assert n3 == 8 * 40 + 2 * 8 and n4 == 8 * 33 + 2 * 32
    assert spans["first_entry"].tolist()[:1] == [0] and spans["first_entry"][3] == n0 and spans["first_entry"][4] == n0 + n3
    assert len(unit) == n0 + n3 + n4
    # variant-major: 8 rows of patterns (cursors 4v into the pattern table) over all planned units, then the message's own
    # last two bits / last bit over the units from tail_first on (no unit in front of them can get that far)
    e0 = unit[:8 * reach0].reshape(8, reach0)
    assert (e0 == np.arange(reach0)[None, :]).all()
    assert cursor[:8 * reach0].reshape(8, reach0)[:, 0].tolist() == [0, 4, 8, 12, 16, 20, 24, 28]
    e0t = unit[8 * reach0:n0].reshape(2, reach0 - t0)
    assert (e0t == np.arange(t0, reach0)[None, :]).all()
    assert cursor[8 * reach0:n0].reshape(2, reach0 - t0)[:, 0].tolist() == [130, 131]
    e3 = unit[n0:n0 + 8 * 40].reshape(8, 40)
    assert (e3 == 9000 * 4 + np.arange(40)[None, :]).all()          # the whole 10-frame stream: 40 units < the message's reach
    # capacity: what does not fit any more is left out, later streams that fit are still planned
    spans2, unit2, _ = mlib.select_plan(segs, n0 + n3 + 10)
    assert spans2["reach"].tolist() == [reach0, 0, 0, 40, 0] and len(unit2) == n0 + n3
    # a message array without the patterns in front cannot be planned
    segs["hide_base"][0] = 0
    assert mlib.select_plan(segs, 1 << 20)[0]["reach"][0] == 0


def test_plan_reach_limit(mlib):
    segs = np.zeros(1, dtype=mlib.CHAIN_SEG_DTYPE)
    segs["n_frames"], segs["hide_base"], segs["hide_begin"] = 100000, 32, 32
    # bits * 5 // 14 + 32 <= MP3S_SELECT_MAX_REACH = 1 << 18 (the selection works in rounds: LDS does not limit the reach)
    for bits, planned in ((5000, True), (5648, True), (50000, True), (733916, True), (733919, False)):
        segs["hide_end"] = 32 + bits
        reach = int(mlib.select_plan(segs, 1 << 22)[0]["reach"][0])
        assert (reach > 0) == planned and reach <= 1 << 18, (bits, reach)
        if planned:
            assert reach == bits * 5 // 14 + 32
    # ... and never past the end of the stream
    segs["n_frames"], segs["hide_end"] = 1000, 32 + 50000
    assert int(mlib.select_plan(segs, 1 << 22)[0]["reach"][0]) == 4000


# ------------------------------------------------------------------------------------------------ the selection (GPU)
@pytest.mark.gpu
def test_short_messages_every_ending(ctx, mlib, orc):
    """one to a few bits (only the "bits left" variants and the start), and messages ending on every offset inside a unit"""
    from synth_pcm import synth_pcm
    pcm = synth_pcm(60, seed=31)
    rng = np.random.default_rng(8)
    for nbits in list(range(1, 14)) + [31, 32, 33, 95, 96, 97, 200, 201, 202, 203, 204, 205]:
        msg = rng.integers(0, 2, size=nbits).astype(np.uint8)
        r = ctx.encode_pcm(pcm, 44100, 128, msg)
        o = orc.encode(pcm, 44100, 128, msg)
        assert o["rc"] == 0 and r["mp3"] == o["mp3"], nbits
        assert r["hide_offset"] == o["hide_offset"] and r["too_long"] == bool(o["too_long"]), nbits
        assert r["rate_passes"] == 1, nbits                          # decided on the device: the first pass was final


@pytest.mark.gpu
def test_long_messages_in_rounds(ctx, mlib, orc):
    """messages whose reach takes several rounds of the selection (1 024 units each, the state carried from round to
    round): final after the first pass; silences that one round can absorb; a message that fills the stream"""
    from synth_pcm import synth_pcm
    pcm = synth_pcm(900, seed=35)
    pcm[150 * 1152:155 * 1152] = 0                                    # 20 silent units in round 0, 24 in round 2
    pcm[600 * 1152:606 * 1152] = 0
    rng = np.random.default_rng(12)
    for nbits in (2900, 3100, 6000, 9001, 20000):
        msg = rng.integers(0, 2, size=nbits).astype(np.uint8)
        r, o = ctx.encode_pcm(pcm, 44100, 128, msg), orc.encode(pcm, 44100, 128, msg)
        assert o["rc"] == 0 and r["mp3"] == o["mp3"], nbits
        assert r["hide_offset"] == o["hide_offset"] and r["too_long"] == bool(o["too_long"]), nbits
        assert r["rate_passes"] == 1, nbits


@pytest.mark.gpu
def test_inherited_addresses_are_put_right_on_the_device(ctx, mlib, orc):
    """silences in the middle of a stream: the granules behind them read address1/2/3 inherited across the silence
    (SURVEY E7); the first pass gives every unit zeros, the chain check lists the units for which that was wrong and they
    run again on the device -- final without the host, same bytes as the oracle"""
    from synth_pcm import synth_pcm
    rng = np.random.default_rng(14)
    for seed, cuts in ((41, [(60, 90)]), (42, [(10, 12), (100, 140), (200, 201)]), (43, [(0, 30), (150, 155)])):
        pcm = synth_pcm(260, seed=seed)
        for a, b in cuts:
            pcm[a * 1152:b * 1152] = 0
        pcm[220 * 1152:230 * 1152, 0] = 0                             # one channel only
        for nbits in (0, 40, 1500):
            msg = rng.integers(0, 2, size=nbits).astype(np.uint8) if nbits else None
            r, o = ctx.encode_pcm(pcm, 44100, 128, msg), orc.encode(pcm, 44100, 128, msg)
            assert o["rc"] == 0 and r["mp3"] == o["mp3"] and r["hide_offset"] == o["hide_offset"], (seed, nbits)
            assert r["rate_passes"] == 1, (seed, nbits)


@pytest.mark.gpu
def test_silence_inside_the_reach_and_uncovered_start(ctx, mlib, orc):
    from synth_pcm import synth_pcm
    rng = np.random.default_rng(9)
    msg = rng.integers(0, 2, size=900).astype(np.uint8)
    # silent frames in the middle of the message: those units take no tables, d grows by three per unit
    pcm = synth_pcm(200, seed=32)
    pcm[20 * 1152:27 * 1152] = 0
    r, o = ctx.encode_pcm(pcm, 44100, 128, msg), orc.encode(pcm, 44100, 128, msg)
    assert r["mp3"] == o["mp3"] and r["hide_offset"] == o["hide_offset"]
    # a long silent start (raw PCM: no input stream whose tables could tell): the message reaches further than planned;
    # the chain check on the device says so, the units behind the plan run again on the cursors it found (300 of them:
    # the list holds 1 024) and the second check passes; without the device's re-runs the host resolves the chains
    # (more than one pass); same bytes every time
    pcm = synth_pcm(300, seed=33)
    pcm[:120 * 1152] = 0
    r, o = ctx.encode_pcm(pcm, 44100, 128, msg), orc.encode(pcm, 44100, 128, msg)
    assert r["mp3"] == o["mp3"] and r["hide_offset"] == o["hide_offset"]
    assert r["rate_passes"] == 1
    keep = ctx.set_option("redo", 0)
    try:
        r = ctx.encode_pcm(pcm, 44100, 128, msg)
    finally:
        ctx.set_option("redo", keep)
    assert r["mp3"] == o["mp3"] and r["hide_offset"] == o["hide_offset"] and r["rate_passes"] > 1
    # ... and a message far behind the plan: 2 100 units to run again (the list holds 4 096 since round 4; 1 024 before, and the
    # host took this one)
    long_msg = rng.integers(0, 2, size=6000).astype(np.uint8)
    pcm = synth_pcm(1500, seed=36)
    pcm[:700 * 1152] = 0
    r, o = ctx.encode_pcm(pcm, 44100, 128, long_msg), orc.encode(pcm, 44100, 128, long_msg)
    assert r["mp3"] == o["mp3"] and r["hide_offset"] == o["hide_offset"]
    # more units than the list holds: the host again
    longer = rng.integers(0, 2, size=14000).astype(np.uint8)
    pcm = synth_pcm(2600, seed=37)
    pcm[:700 * 1152] = 0
    r, o = ctx.encode_pcm(pcm, 44100, 128, longer), orc.encode(pcm, 44100, 128, longer)
    assert r["mp3"] == o["mp3"] and r["hide_offset"] == o["hide_offset"] and r["rate_passes"] > 1
    # a stream shorter than the message's reach, message longer than the stream can hold
    pcm = synth_pcm(9, seed=34)
    r, o = ctx.encode_pcm(pcm, 44100, 128, msg), orc.encode(pcm, 44100, 128, msg)
    assert r["mp3"] == o["mp3"] and r["too_long"] and bool(o["too_long"]) and r["hide_offset"] == o["hide_offset"]


@pytest.mark.gpu
def test_batch_of_streams_and_blocks(ctx, mlib, orc):
    """several streams with their own plans in one batch (one workgroup per stream), and a message carried across blocks
    (the plan starts at the carried cursor)"""
    from synth_pcm import synth_pcm
    files, msgs = [], []
    for i, n in enumerate((90, 40, 300, 55)):
        pcm = synth_pcm(n, seed=50 + i)
        if i == 2:
            pcm[10 * 1152:14 * 1152] = 0
        files.append(bytes(ctx.encode_pcm(pcm, 44100, 128, None)["mp3"]))
        msgs.append("stream %d " % i * (1 + 3 * i))
    msgs[1] = None                                                    # one of them is cleared
    got = ctx.hide_messages(files, msgs)
    for i, (f, m) in enumerate(zip(files, msgs)):
        one = ctx.hide_message(f, m) if m is not None else ctx.clear_file(f)
        assert bytes(got[i]["data"]) == bytes(one["data"]) and got[i]["hide_offset"] == one["hide_offset"], i
        d = orc.decode(f)
        o = orc.encode(orc.pcm_to_i16(d["pcm"]), 44100, 128, np.array(mlib.message_frame(m), dtype=np.uint8) if m is not None else None)
        assert bytes(got[i]["data"]) == o["mp3"], i
    # blocks: 3 x 50 frames, a message that runs from the first block into the second
    pcm = synth_pcm(150, seed=60)
    msg = np.random.default_rng(10).integers(0, 2, size=700).astype(np.uint8)
    whole = ctx.encode_pcm(pcm, 44100, 128, msg)
    out, carry = b"", None
    for b in range(3):
        lead = 0 if b == 0 else 2
        blk = ctx.encode_block(pcm[(b * 50 - lead) * 1152:(b + 1) * 50 * 1152], lead, b * 50, b == 2, 44100, 128, msg, carry)
        out += bytes(blk["mp3"])
        carry = blk["carry_out"]
    assert out == whole["mp3"] == orc.encode(pcm, 44100, 128, msg)["mp3"]


@pytest.mark.gpu
def test_rate_select_entry_point(ctx, mlib):
    """mp3s_rate_select_dev + mp3s_chain_resolve_dev on device buffers, two streams: GrInfo, quantised lines and the
    cursors equal what the library's own encoder (pinned to the oracle above) leaves for the same streams"""
    from synth_pcm import synth_pcm
    L = mlib.lib()
    n0, n1 = 120, 70
    n, units = n0 + n1, (n0 + n1) * 4
    pcm0, pcm1 = synth_pcm(n0, seed=71), synth_pcm(n1, seed=72)
    pcm1[5 * 1152:7 * 1152] = 0
    rng = np.random.default_rng(11)
    m0, m1 = rng.integers(0, 2, size=333).astype(np.uint8), rng.integers(0, 2, size=64).astype(np.uint8)
    want = [ctx.encode_pcm(pcm0, 44100, 128, m0), ctx.encode_pcm(pcm1, 44100, 128, m1)]
    hdr = np.zeros(n, dtype=mlib.FRAME_HDR_DTYPE)
    hdr["nch"] = 2
    hdr["stream_first"][n0:] = n0
    mdct = ctx.encode_transform(np.concatenate([pcm0, pcm1]), hdr)
    rf = np.concatenate([mlib.rate_frames(44100, 128, 2, n0)[0], mlib.rate_frames(44100, 128, 2, n1)[0]])
    hide = np.concatenate([mlib.select_patterns(), m0, m1])
    rf["stream"][n0:] = 1
    rf["hide_end"][:n0] = 32 + len(m0)
    rf["hide_end"][n0:] = 32 + len(m0) + len(m1)
    segs = np.zeros(2, dtype=mlib.CHAIN_SEG_DTYPE)
    segs["first_frame"], segs["n_frames"] = [0, n0], [n0, n1]
    segs["hide_base"] = segs["hide_begin"] = [32, 32 + len(m0)]
    segs["hide_end"] = [32 + len(m0), 32 + len(m0) + len(m1)]
    spans, eu, ec = mlib.select_plan(segs, 1 << 20)
    ne, max_reach = len(eu), int(spans["reach"].max())
    assert spans["reach"].min() > 0
    cur = np.full(units, mlib.NO_CURSOR, dtype=np.int32)
    dev = [ctx.to_device(a) for a in (mdct, rf, hide, cur, segs, spans, eu, ec)]
    d_mdct, d_rf, d_hide, d_cur, d_segs, d_spans, d_eu, d_ec = dev
    outs = [ctx.alloc(n * 2304 * 2), ctx.alloc(units * 72), ctx.alloc(units * 88), ctx.alloc(ne * 1152),
            ctx.alloc(ne * 72 + ((ne + 15) & ~15)), ctx.alloc(ne * 88), ctx.alloc(16), ctx.alloc(2 * 80)]
    d_ix, d_out, d_en, d_ixv, d_outv, d_env, d_ver, d_so = outs
    try:
        ctx2 = mlib.Context(ctx.device)
        for it in range(3):  # the second call finds the cursors the first one wrote (exact ones): same result
            if it < 2:
                mlib.check(L.mp3s_rate_select_dev(ctx.handle, d_mdct, d_rf, n, d_hide, len(hide), d_cur, d_segs, d_spans, 2, max_reach,
                                                  d_eu, d_ec, ne, d_ix, d_out, d_en, d_ixv, d_outv, d_env))
                mlib.check(L.mp3s_chain_resolve_dev(ctx.handle, d_out, d_rf, n, d_segs, 2, d_cur, None, d_ver, d_so))
            else:
                # the third time in two halves: the rate loop on this context, selection and chain check on another one's
                # stream (what bench.py does with the tail of a batch)
                mlib.check(L.mp3s_rate_variants_dev(ctx.handle, d_mdct, d_rf, n, d_hide, len(hide), d_cur, d_eu, d_ec, ne, d_ix, d_out,
                                                    d_en, d_ixv, d_outv, d_env))
                ctx2.wait_for(ctx)
                mlib.check(L.mp3s_select_dev(ctx2.handle, d_hide, d_cur, d_segs, d_spans, 2, max_reach, d_eu, d_ec, ne, d_ix, d_out, d_en,
                                             d_ixv, d_outv, d_env))
                mlib.check(L.mp3s_chain_resolve_dev(ctx2.handle, d_out, d_rf, n, d_segs, 2, d_cur, None, d_ver, d_so))
                ctx.wait_for(ctx2)
            gr = ctx.download(d_out, mlib.GR_OUT_DTYPE, (units,))
            ver = ctx.download(d_ver, np.int32, (2,))
            so = ctx.download(d_so, mlib.CHAIN_SEG_OUT_DTYPE, (2,))
            used = ctx.download(d_cur, np.int32, (units,))
            assert ver.tolist() == [0, 0]
            for s, (a, b) in enumerate(((0, n0 * 4), (n0 * 4, units))):
                w = want[s]["gr"]
                for k in ("part2_3_length", "big_values", "count1", "table_select", "count1table_select", "region0_count",
                          "region1_count", "n_tables", "quantizer_step", "address"):
                    assert np.array_equal(gr[k][a:b], w[k]), (s, k)
                base = int(segs["hide_base"][s])
                assert int(so["cursor"][s]) - base == want[s]["hide_offset"]
                # the cursors the units really saw: the running sum of the tables, up to the end of the message
                true = base + np.concatenate([[0], np.cumsum(w["n_tables"])[:-1]])
                hiding = true < int(segs["hide_end"][s])
                assert np.array_equal(used[a:b][hiding], true[hiding]) and (used[a:b][~hiding] == mlib.NO_CURSOR).all(), s
    finally:
        for p in dev + outs:
            ctx.free(p)
        if "ctx2" in locals():
            ctx2.close()


@pytest.mark.gpu
def test_chains_of_inheriting_units_are_followed_on_the_device(ctx, mlib, orc):
    """behind a silence, granules so quiet that they take no big values keep the addresses they inherit and hand them on
    (encoder/MP3_Encoder.py:1004-1006, 788-803): the unit behind such a unit inherits what THAT one was given.  The
    first pass gives everybody zeros; the check can only fault the first of the row (the others were given what their
    predecessor -- wrongly -- left); its re-run then follows the chain from unit to unit (k_rate_redo, round 4) where one re-run
    per check used to leave the rest to the host.  Same bytes as the oracle, final after the first pass."""
    from synth_pcm import synth_pcm
    rng = np.random.default_rng(23)
    passes = []
    for seed, quiet_frames, amp in ((51, 6, 1), (52, 12, 2), (53, 25, 1), (54, 40, 3)):
        pcm = synth_pcm(160, seed=seed)
        pcm[40 * 1152:60 * 1152] = 0                                  # silence ...
        q = rng.integers(-amp, amp + 1, size=(quiet_frames * 1152, 2)).astype(np.int16)
        pcm[60 * 1152:(60 + quiet_frames) * 1152] = q                 # ... then a row of frames that are active but take no (or hardly any) big values
        for nbits in (0, 64):
            msg = rng.integers(0, 2, size=nbits).astype(np.uint8) if nbits else None
            r, o = ctx.encode_pcm(pcm, 44100, 128, msg), orc.encode(pcm, 44100, 128, msg)
            assert o["rc"] == 0 and r["mp3"] == o["mp3"] and r["hide_offset"] == o["hide_offset"], (seed, nbits)
            passes.append(r["rate_passes"])
    assert passes == [1] * len(passes), passes
