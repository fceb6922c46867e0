#!/usr/bin/env python3
"""Extract Mercury's LDPC parity-check graphs into a compact binary (derived DATA, not source).

Runs only in the build container: it loads oracle/_ref/libmercury_ref.so (the reference compiled
from /root/reference by oracle/Makefile) and reads the table globals selected by
source/physical_layer/ldpc.cc:140-251 (mercury_normal_{1,2,3,4,5,6,8,14}_16.cc).

Only what cannot be derived is stored: the check->variable adjacency in the reference's row
order (QCmatrixC) and the variable->check adjacency in the reference's SLOT order (QCmatrixV; the
slot order fixes the summation order of the sum-product variable update,
ldpc_decoder_SPA.cc:162-170). The script asserts that everything else the reference ships is
derivable: -1 padding strictly trailing, QCmatrixd == run-length of the V-row degrees,
QCmatrixEnc[i] == QCmatrixC[i] minus the check's own parity bit K+i (same order).

File layout (little endian):
  u32 magic 'MLDP', u32 version=1, u32 nrates
  per rate: u32 K, P, N, E, Cwidth, Vwidth ; u8 cdeg[P] ; u16 C[E] ; u8 vdeg[N] ; u16 V[E]
"""
import ctypes, struct, sys, os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "oracle/_ref/libmercury_ref.so"))
L.mref_create.restype = ctypes.c_void_p


class Info(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                "cfg M bits_per_symbol K P N Nsymb Nc Nfft Ngi Nofdm nData nBits nPilots nVirtual nReal "
                "bit_blk tf_blk preamble_nsymb estimator amp_restore ls_window Cwidth Vwidth dwidth payload_bytes".split()]


def main(out_path):
    blobs = []
    # one cfg per distinct rate: 1,2,3,4,5,6,8,14 /16
    for cfg in [0, 1, 2, 3, 4, 5, 6, 12]:
        h = ctypes.c_void_p(L.mref_create(cfg, 50))
        i = Info()
        L.mref_get_info(h, ctypes.byref(i))
        P, N, K, cw, vw, dw = i.P, i.N, i.K, i.Cwidth, i.Vwidth, i.dwidth
        C = np.zeros((P, cw), np.int32)
        V = np.zeros((N, vw), np.int32)
        d = np.zeros(dw, np.int32)
        Enc = np.zeros((P, cw - 1), np.int32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        L.mref_get_ldpc_tables(h, p(C), p(V), p(d), p(Enc))
        cdeg = (C >= 0).sum(1)
        vdeg = (V >= 0).sum(1)
        E = int(cdeg.sum())
        assert E == int(vdeg.sum())
        for row, n in zip(C, cdeg):
            assert (row[:n] >= 0).all() and (row[n:] < 0).all()
        for row, n in zip(V, vdeg):
            assert (row[:n] >= 0).all() and (row[n:] < 0).all()
        adj = [set() for _ in range(N)]
        for c in range(P):
            for j in range(cdeg[c]):
                adj[C[c, j]].add(c)
        for v in range(N):
            assert set(V[v, :vdeg[v]].tolist()) == adj[v] and len(adj[v]) == vdeg[v]
        dd = []
        for s in range(0, dw, 2):
            dd += [d[s + 1]] * d[s]
        assert len(dd) == N and all(dd[v] == vdeg[v] for v in range(N))
        encdeg = (Enc >= 0).sum(1)
        for c in range(P):
            assert Enc[c, :encdeg[c]].tolist() == [x for x in C[c, :cdeg[c]].tolist() if x != K + c]
            assert (K + c) in C[c, :cdeg[c]].tolist()
        Cf = np.concatenate([C[c, :cdeg[c]] for c in range(P)]).astype("<u2")
        Vf = np.concatenate([V[v, :vdeg[v]] for v in range(N)]).astype("<u2")
        blobs.append(struct.pack("<6I", K, P, N, E, cw, vw) + cdeg.astype("u1").tobytes() + Cf.tobytes()
                     + vdeg.astype("u1").tobytes() + Vf.tobytes())
        print(f"rate {K}/1600: P={P} E={E} Cwidth={cw} Vwidth={vw}")
    with open(out_path, "wb") as f:
        f.write(struct.pack("<4sII", b"MLDP", 1, len(blobs)))
        for b in blobs:
            f.write(b)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "mercury_amd/data/mercury_ldpc_tables.bin"))
