"""Randomised sweep of the pigpio-rule decoders on the GPU: somfy / berner / mumbi / kwmobile / baldr symbol
streams (random payloads, several transmissions, junk bursts, idle gaps, truncated repeats) at random sample
rates, run through EVERY rule decoder (sdr_process_bits); frames compared with the oracle sniffers fed from
the same edges.  usage: python tools/fuzz_rules.py [cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import decoders as odec
from sdr_amd import chain, synth

RULES = [(chain.DEC_SOMFY, odec.SomfySniffer, 20000), (chain.DEC_BERNER, odec.BernerSniffer, None),
         (chain.DEC_MUMBI, odec.MumbiSniffer, 8000), (chain.DEC_KWMOBILE, odec.KwmobileDecoder, 3000),
         (chain.DEC_BALDR, odec.BaldrDecoder, 18000)]


NFRAMES = [0]


def emitter(rng):
    k = int(rng.integers(0, 5))
    if k == 0:
        fr = synth.somfy_frame(0xA0 | int(rng.integers(0, 16)), int(rng.integers(1, 9)), int(rng.integers(0, 65536)),
                               int(rng.integers(0, 1 << 24)))
        return synth.somfy_symbols(fr, repeats=int(rng.integers(0, 3)), wakeup=bool(rng.integers(0, 2)))
    if k == 1:
        return synth.berner_symbols(int(rng.integers(0, 1 << 32)), repeats=int(rng.integers(1, 6)))
    if k == 2:
        codes = sorted(synth.MUMBI_CODES)
        code = bytes(bytearray(synth.MUMBI_CODES[codes[int(rng.integers(0, len(codes)))]]))
        return synth.mumbi_symbols(code, repeats=int(rng.integers(1, 6)), on_air=bool(rng.integers(0, 2)))
    if k == 3:
        bits = synth.kwmobile_bits(int(rng.integers(0, 256)), int(rng.integers(0, 3)), bool(rng.integers(0, 2)),
                                   int(rng.integers(-300, 500)), int(rng.integers(0, 100)))
        return synth.kwmobile_symbols(bits, repeats=int(rng.integers(2, 8)))
    bits = synth.baldr_bits(int(rng.integers(0, 256)), int(rng.integers(0, 3)), int(rng.integers(0, 2)),
                            int(rng.integers(-300, 500)), int(rng.integers(0, 100)))
    return synth.baldr_symbols(bits, repeats=int(rng.integers(1, 5)))


def one(rng, k):
    rate = int(rng.choice([100000, 200000, 250000, 1000000]))
    sym = []
    for _ in range(int(rng.integers(1, 5))):
        s = list(emitter(rng))
        if rng.random() < 0.2:
            s = s[:int(rng.integers(1, len(s) + 1))]          # transmission cut short
        jit = 1.0 + 0.04 * (rng.random() - 0.5)
        sym += [(lv, w * jit) for lv, w in s]
        r = rng.random()
        if r < 0.3:
            for _ in range(int(rng.integers(1, 30))):
                sym += [(1, float(rng.integers(30, 1200))), (0, float(rng.integers(30, 1200)))]
        elif r < 0.6:
            sym += [(0, float(rng.integers(1000, 40000)))]
    env = synth.envelope(sym, rate, lead_in_us=float(rng.integers(0, 12000))).astype(np.int8)
    env = np.concatenate([env, np.zeros(int(rng.integers(0, 5000)), np.int8)])
    if rng.random() < 0.2:
        env = 1 - env                                         # starts high
    desc = "case %d: rate %d, %d bytes" % (k, rate, env.size)
    with chain.Chain(rate, decoder=chain.DEC_SOMFY, decoder_rate=rate, max_samples=env.size, max_frames=64) as c:
        for j, (dec, cls, wd) in enumerate(RULES):
            if j == 0:
                c.process_bits(env)
            else:
                c.redecode(dec)
            frames = c.fetch_frames()
            # the same stream in ragged calls on a streaming handle (state carried between calls)
            cuts = sorted(set([0, env.size] + [int(v) for v in rng.integers(0, env.size, int(rng.integers(1, 12)))]))
            with chain.Chain(rate, decoder=dec, decoder_rate=rate, max_samples=env.size, max_frames=64) as cs:
                chunked = []
                for a0, b0 in zip(cuts[:-1], cuts[1:]):
                    cs.process_bits(env[a0:b0])
                    chunked += [(f.packet(), bytes(bytearray(f.raw[:7])), f.status, f.nbits) for f in cs.fetch_frames()]
            if chunked != [(f.packet(), bytes(bytearray(f.raw[:7])), f.status, f.nbits) for f in frames]:
                return desc + " decoder %d: %d-call streaming pass differs from the one-call pass (cuts %s)" % (dec, len(cuts) - 1, cuts)
            s = cls()
            for lvl, tick in odec.edges_to_events(env, rate, watchdog_us=wd):
                s.cbf(lvl, tick)
            if len(frames) != len(s.frames):
                return desc + " decoder %d: %d frames vs oracle %d" % (dec, len(frames), len(s.frames))
            NFRAMES[0] += len(frames)
            for a, b in zip(s.frames, frames):
                if dec == chain.DEC_SOMFY:
                    ok = b.packet() == a["frame"] and bytes(bytearray(b.raw[:7])) == a["raw"]
                elif dec in (chain.DEC_KWMOBILE, chain.DEC_BALDR):
                    ok = b.packet() == a["frame"] and b.temp_tenths / 10. == a["temperature"] and \
                        (b.v[0], b.v[1], b.v[3]) == (a["sensor_id"], a["humidity"], a["channel"])
                else:
                    ok = b.packet() == a
                if not ok:
                    return desc + " decoder %d: frame differs: %r vs %r" % (dec, b.packet(), a)
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
    print("done: %d cases, %d failures, %d frames compared" % (cases, fails, NFRAMES[0]))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
