"""Golden vectors for the text dumpers of gaba_parse.h (extended CIGAR, gapped sequence rows) and gaba_dp_calc_score: random alignments traced by the
*compiled reference* (oracle/_ref/libgaba_ref.so), each segment pushed through the reference's own dumpers (ref_harness/gaba_ref_shim.c:shim_dumpers).
Run in the build container:  python tests/golden/make_dumper_golden.py"""
import ctypes, gzip, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, gabalib as G

HERE = os.path.dirname(os.path.abspath(__file__))
CAP = 1 << 15
MODELS = [('pacbio', G.PACBIO), ('ont1dsq', G.ONT1DSQ), ('affine', G.AFFINE_DEFAULT)]

def padded(x):
    p = np.full(len(x) + 128, 4, dtype=np.uint8); p[64:64 + len(x)] = x; return p

def path_buffer(words, plen):
    """the parsers peek below path[0]: keep the {plen, 0x40000000} header of gaba_alignment_s in front, zeros behind"""
    arr = (ctypes.c_uint32 * (len(words) + 10))(); arr[0] = plen; arr[1] = 0x40000000
    for i, w in enumerate(words): arr[2 + i] = w
    return arr, ctypes.addressof(arr) + 8

def main():
    out = []
    for mname, model in MODELS:
        ref = G.Reference(**model)
        jobs = G.random_jobs(7101 + len(mname), 60, max_len=1500)
        for (a, apos, arev, b, bpos, brev, bw, _) in jobs:
            d = ref.extend(a, apos, arev, b, bpos, brev, bw, 1)
            if d['traced'] != 1: continue
            pa, pb = padded(a), padded(b)
            arr, base = path_buffer(d['path'], d['plen'])
            for s in d['segs']:
                if s[0] > 1 or s[1] < 2 or s[1] > 3 or s[4] + s[5] == 0: continue          # sections of the 96 x N tail: not part of the two sequences
                seg = G.Seg(*s)
                buf = ctypes.create_string_buffer(5 * CAP); sc = (ctypes.c_int64 * 12)(); idt = ctypes.c_double()
                rc = ref.L.shim_dumpers(ref.dp, ctypes.c_void_p(pa.ctypes.data + 64), len(a), int(arev), ctypes.c_void_p(pb.ctypes.data + 64), len(b), int(brev),
                                        ctypes.c_void_p(base), ctypes.byref(seg), buf, ctypes.c_uint64(CAP), sc, ctypes.byref(idt))
                assert rc == 0
                strs = [ctypes.string_at(ctypes.addressof(buf) + i * CAP).decode() for i in range(5)]
                out.append(dict(model=mname, a=''.join(map(str, a)), arev=int(arev), b=''.join(map(str, b)), brev=int(brev), plen=d['plen'], path=d['path'], seg=list(s),
                                xcigar_f=strs[0], xcigar_r=strs[1], row_a=strs[2], row_b=strs[3], row_a_rev=strs[4], score=[int(x) for x in sc],
                                identity=np.float64(idt.value).tobytes().hex()))
    with gzip.GzipFile(os.path.join(HERE, 'gaba_dumpers.json.gz'), 'wb', mtime=0) as f: f.write(json.dumps(out).encode())
    print(len(out), 'segments;', sum(1 for o in out if o['score'][11] != 0), 'with a nonzero adj;', sum(1 for o in out if 'X' in o['xcigar_f']), 'with mismatches')

if __name__ == '__main__':
    main()
