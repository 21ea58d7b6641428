"""Randomised decoder sweep on the GPU: TS33C / TFA symbol streams with jitter, junk bursts and idle gaps
at random sample rates (rows from a few KiB to > 1 MiB: one-round, multi-round and segment-parallel
run-length paths), cut into random chunks; frames and decoder state after EVERY call compared with the
oracle StreamDecoder.  usage: python tools/fuzz_decode.py [cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import decoders as odec
from sdr_amd import chain, synth
from tests.test_host_decode import assert_tfa_equal, assert_ts33c_equal


def stream(rng, tfa, rate):
    sym = []
    for _ in range(int(rng.integers(1, 7))):
        if tfa:
            p = synth.tfa_packet(ident=int(rng.integers(0, 65536)), temp_c=float(rng.integers(-300, 500)) / 10,
                                 humidity=int(rng.integers(0, 100)), battery_ok=bool(rng.integers(0, 2)))
            sym += synth.tfa_symbols(synth.tfa_bits(p), rng, reset_gap_us=float(rng.integers(600, 9000)))
        else:
            p = synth.ts33c_packet(**synth.random_ts33c_fields(rng))
            sym += synth.ts33c_symbols(synth.ts33c_bits(p), rng)
        r = rng.random()
        if r < 0.25:      # junk burst: random short pulses
            for _ in range(int(rng.integers(1, 40))):
                sym += [(1, float(rng.integers(20, 900))), (0, float(rng.integers(20, 900)))]
        elif r < 0.5:
            sym += [(0, float(rng.integers(2000, 60000)))]
    env = synth.envelope(sym, rate, lead_in_us=float(rng.integers(0, 8000)))
    return np.concatenate([env, np.zeros(int(rng.integers(0, 3000)), np.uint8)]).astype(np.int8)


def one(rng, k):
    tfa = rng.random() < 0.5
    rate = int(rng.choice([500000, 1000000, 2000000])) if tfa else int(rng.choice([20000, 50000, 250000, 1000000]))
    env = stream(rng, tfa, rate)
    desc = "case %d: %s rate %d, %d bytes" % (k, "TFA" if tfa else "TS33C", rate, env.size)
    grc = tfa and rng.random() < 0.5
    o = (odec.TFAStreamDecoder(sample_rate=rate, nan_guard=grc) if tfa else odec.TS33CStreamDecoder(sample_rate=rate))
    dec = (chain.DEC_TFA_GRC if grc else chain.DEC_TFA) if tfa else chain.DEC_TS33C
    ncut = int(rng.choice([0, 1, 3, 12]))
    cuts = np.sort(rng.integers(0, env.size + 1, ncut))
    frames = []
    with chain.Chain(rate, decoder=dec, decoder_rate=rate, max_samples=env.size, max_frames=2048) as c:
        for a, e in zip(np.r_[0, cuts], np.r_[cuts, env.size]):
            ch = env[a:e]
            try:
                o.work([ch])
                oerr = None
            except (ValueError, IndexError) as ex:   # the reference's latent exceptions
                oerr = type(ex).__name__
            c.process_bits(ch)
            got = c.fetch_frames()
            if oerr:
                return None   # the oracle object is dead after an exception; statuses are covered by the unit tests
            frames += got
            st = [int(o.state == "frame"), int(o.currentSymbols.size), int(o.previousSymbolType), o.chunk_id]
            if c.decoder_state() != st:
                return desc + " state %s vs oracle %s after chunk [%d,%d)" % (c.decoder_state(), st, a, e)
    if len(frames) != len(o.frames):
        return desc + " %d frames vs oracle %d" % (len(frames), len(o.frames))
    for of, gf in zip(o.frames, frames):
        try:
            (assert_tfa_equal if tfa else assert_ts33c_equal)(of, gf)
        except AssertionError as ex:
            return desc + " frame differs: %r" % (ex,)
    return None


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0, fails = time.time(), 0
    for k in range(cases):
        r = one(rng, k)
        if r:
            fails += 1
            print("FAIL", r, flush=True)
        if k % 20 == 19:
            print("%d cases, %d failures, %.0f s" % (k + 1, fails, time.time() - t0), flush=True)
    print("done: %d cases, %d failures" % (cases, fails))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
