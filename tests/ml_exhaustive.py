"""Brute-force maximum-likelihood reference for SHORT blocks of the trellis: every input sequence is encoded, the one
closest to the received values wins, and among equally close ones the one the tie rule names.

The tie rule as STATED in oracle/tetra_oracle.c (what libosmocore's two decoders do, lower_mac/viterbi_cch.c:58-66 ->
osmo_conv_decode): where two paths merge with equal metric, the survivor is the predecessor t >> 1 -- the path whose
oldest state bit is 0.  Unrolled from the end of the block (start state 0, L steps, 4 flush steps with input 0): among all
sequences of minimum distance the decoder returns the one that is smallest when read from the LAST bit to the first, 0
before 1.  This module is that sentence and nothing else; the tests hold both restated algorithms (and the kernels) to it
on every received word of a short block, so a single run against libosmocore later pins the rule for everything at once."""
import numpy as np

import oraclelib as O


def codebook(L, K, mother, pu):
    """(2^L, K) array: the punctured code word of every L-bit input (the encoder starts in state 0; no tail is forced)"""
    xs = ((np.arange(1 << L)[:, None] >> np.arange(L)[None, :]) & 1).astype(np.uint8)        # xs[v, i] = bit i of v
    return xs, np.stack([O.conv_encode_block(pu, mother, x, K) for x in xs])


def ml_decode(xs, cb, rx):
    """rx: (n, K) received values 0 / 1 / 0xff (erased).  Returns (decoded inputs (n, L), minimum distances, number of
    inputs at that distance)"""
    n = len(rx)
    out = np.zeros((n, xs.shape[1]), np.uint8)
    dmin = np.zeros(n, np.int64)
    nties = np.zeros(n, np.int64)
    known = rx != 0xff
    for i in range(n):
        d = ((cb != rx[i][None, :]) & known[i][None, :]).sum(1)
        m = d.min()
        best = np.flatnonzero(d == m)
        # value of the sequence read from the last bit to the first = the index v itself (bit i of v = input i): smallest wins
        out[i] = xs[best.min()]
        dmin[i], nties[i] = m, len(best)
    return out, dmin, nties


def erasure_patterns(K, rng, extra=6):
    """none, every single position, the first / second half, every other position, and a few random ones"""
    pats = [np.zeros(K, bool)]
    for k in range(K):
        p = np.zeros(K, bool); p[k] = True; pats.append(p)
    p = np.zeros(K, bool); p[:K // 2] = True; pats.append(p)
    p = np.zeros(K, bool); p[K // 2:] = True; pats.append(p)
    p = np.zeros(K, bool); p[::2] = True; pats.append(p)
    for _ in range(extra):
        pats.append(rng.random(K) < 0.3)
    return pats
