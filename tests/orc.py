"""ctypes view of oracle/liboracle.so -- the CHECKER used by the tests (never by the product)."""
import ctypes as C
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_L = None


def lib():
    global _L
    if _L is None:
        L = C.CDLL(os.path.join(REPO, "oracle", "liboracle.so"))
        vp, i = C.c_void_p, C.c_int
        L.orc_build.restype = vp; L.orc_build.argtypes = [C.c_char_p, i, i]
        L.orc_build_ext.restype = vp; L.orc_build_ext.argtypes = [C.c_char_p, i, i]
        L.orc_error.restype = C.c_char_p; L.orc_error.argtypes = [vp]
        L.orc_free.argtypes = [vp]; L.orc_free.restype = None
        L.orc_ffdm.argtypes = [vp, i, i]
        for f in ("orc_num_chunks", "orc_num_patterns", "orc_max_len"):
            getattr(L, f).argtypes = [vp]
        for f in ("orc_state_num", "orc_final_num", "orc_chunk_max_len"):
            getattr(L, f).argtypes = [vp, i]
        L.orc_trie_row.restype = C.POINTER(C.c_int); L.orc_trie_row.argtypes = [vp, i, i]
        L.orc_idmap.restype = C.POINTER(C.c_int); L.orc_idmap.argtypes = [vp, i]
        L.orc_phf_stat.argtypes = [vp, i, i]
        L.orc_phf_lookup.argtypes = [vp, i, i, i]
        for f in ("orc_phf_r", "orc_phf_HT", "orc_phf_val"):
            getattr(L, f).restype = C.POINTER(C.c_int); getattr(L, f).argtypes = [vp, i]
        L.orc_matches_new.restype = vp
        L.orc_matches_free.argtypes = [vp]; L.orc_matches_free.restype = None
        L.orc_matches_count.restype = C.c_int64; L.orc_matches_count.argtypes = [vp]
        L.orc_matches_pos.restype = C.POINTER(C.c_int64); L.orc_matches_pos.argtypes = [vp]
        L.orc_matches_id.restype = C.POINTER(C.c_int32); L.orc_matches_id.argtypes = [vp]
        L.orc_matches_checksum.restype = C.c_uint64; L.orc_matches_checksum.argtypes = [vp]
        L.orc_match_hash.restype = C.c_uint64; L.orc_match_hash.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_scan_reference.argtypes = [vp, vp, C.c_int64, vp]
        L.orc_scan_spec.argtypes = [vp, vp, C.c_int64, vp]
        L.orc_emit.restype = C.c_int64; L.orc_emit.argtypes = [vp, C.c_char_p]
        L.ac_build.restype = vp; L.ac_build.argtypes = [vp]
        L.ac_free.argtypes = [vp]; L.ac_free.restype = None
        L.ac_num_states.argtypes = [vp]
        L.ac_scan_count.restype = C.c_int64; L.ac_scan_count.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_uint64)]
        L.ac_scan_collect.restype = C.c_int64; L.ac_scan_collect.argtypes = [vp, vp, C.c_int64, vp]
        L.ac_scan_count_range.restype = C.c_int64
        L.ac_scan_count_range.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.POINTER(C.c_uint64)]
        _L = L
    return _L


class Oracle:
    """CPU restatement of the reference pipeline for one pattern file."""

    def __init__(self, pattern_file, streamnum=1, gpu_s=4, escapes=False):
        self.L = lib()
        build = self.L.orc_build_ext if escapes else self.L.orc_build
        self.m = build(os.fsencode(pattern_file), streamnum, gpu_s)
        e = self.L.orc_error(self.m)
        if e:
            raise RuntimeError(e.decode())
        self.P = self.L.orc_num_chunks(self.m)
        self.width = None

    def ffdm(self, width, exact=True):
        rc = self.L.orc_ffdm(self.m, width, 1 if exact else 0)
        if rc:
            raise RuntimeError(self.L.orc_error(self.m).decode())
        self.width = width

    def _collect(self, fn, data, n):
        buf = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
        n = buf.size if n is None else n
        o = self.L.orc_matches_new()
        rc = fn(self.m, buf.ctypes.data, n, o)
        if rc:
            self.L.orc_matches_free(o)
            raise RuntimeError(self.L.orc_error(self.m).decode())
        return o

    def _arrays(self, o):
        cnt = self.L.orc_matches_count(o)
        if cnt == 0:
            self.L.orc_matches_free(o)
            return np.empty(0, dtype=np.int64), np.empty(0, dtype=np.int32)
        pos = np.ctypeslib.as_array(self.L.orc_matches_pos(o), (cnt,)).copy()
        ids = np.ctypeslib.as_array(self.L.orc_matches_id(o), (cnt,)).copy()
        self.L.orc_matches_free(o)
        return pos, ids

    def scan_reference(self, data, n=None):
        """tile-faithful restatement of kernel + merge (needs ffdm); returns (pos[], id[]) in output order"""
        return self._arrays(self._collect(self.L.orc_scan_reference, data, n))

    def scan_spec(self, data, n=None):
        """direct dense-trie walk bounded by n; returns (pos[], id[]) in output order"""
        return self._arrays(self._collect(self.L.orc_scan_spec, data, n))

    def emit(self, data, path, n=None, spec=False):
        o = self._collect(self.L.orc_scan_spec if spec else self.L.orc_scan_reference, data, n)
        b = self.L.orc_emit(o, os.fsencode(path))
        cnt = self.L.orc_matches_count(o)
        self.L.orc_matches_free(o)
        return cnt, b

    def trie(self, c=0):
        S = self.L.orc_state_num(self.m, c)
        return np.stack([np.ctypeslib.as_array(self.L.orc_trie_row(self.m, c, s), (256,)) for s in range(S)])

    def stats(self, c=0):
        L = self.L
        return {"state_num": L.orc_state_num(self.m, c), "final": L.orc_final_num(self.m, c),
                "keys": L.orc_phf_stat(self.m, c, 0), "max_key": L.orc_phf_stat(self.m, c, 1),
                "max_offset": L.orc_phf_stat(self.m, c, 2), "r_size": L.orc_phf_stat(self.m, c, 3),
                "ht_size": L.orc_phf_stat(self.m, c, 4)}

    def idmap(self, c=0):
        n = self.L.orc_final_num(self.m, c)
        return np.ctypeslib.as_array(self.L.orc_idmap(self.m, c), (max(n, 1),))[:n].copy()

    def close(self):
        if self.m:
            self.L.orc_free(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def match_checksum(pos, ids):
    """sum of orc_match_hash over (pos, id) pairs, mod 2^64 (numpy twin of oracle/ac_serial.c match_hash)."""
    pos = np.asarray(pos, dtype=np.uint64)
    ids = np.asarray(ids, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (pos + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) ^ (ids * np.uint64(0xC2B2AE3D27D4EB4F))
        x ^= x >> np.uint64(29)
        x = x * np.uint64(0xBF58476D1CE4E5B9)
        return int(x.sum(dtype=np.uint64))


def ac_whole_shard(pattern_file, buf, n_owned=None, base=0, threads=None, slice_bytes=32 << 20):
    """(match count, checksum) of ONE serial Aho-Corasick pass over the whole buffer ``buf`` (numpy uint8, the bytes a
    GPU scan read: n_avail = buf.size), counting the matches that START in [0, n_owned) -- computed on ``threads`` host
    threads, each slice warmed up over the max_pat_len-1 bytes in front of it (oracle/ac_serial.c ac_scan_count_range).
    The checksum is the one pfac_records_checksum computes on the GPU (sum of match_hash(base + pos, pattern id))."""
    import threading
    L = lib()
    o = Oracle(pattern_file, 1, 1)
    ac = L.ac_build(o.m)
    maxlen = L.orc_max_len(o.m)
    n_avail = int(buf.size)
    n_owned = n_avail if n_owned is None else int(n_owned)
    if threads is None:
        try:
            threads = len(os.sched_getaffinity(0))
        except AttributeError:
            threads = os.cpu_count() or 1
        threads = max(1, min(threads, 32))
    cuts = list(range(0, n_avail, slice_bytes)) + [n_avail]
    jobs = list(zip(cuts[:-1], cuts[1:]))          # hits ENDING in [a, b)
    out = [(0, 0)] * len(jobs)
    nxt = [0]
    mu = threading.Lock()
    addr = buf.ctypes.data

    def work():
        while True:
            with mu:
                k = nxt[0]
                nxt[0] += 1
            if k >= len(jobs):
                return
            a, b = jobs[k]
            w = max(0, a - (maxlen - 1))           # cold start here
            chk = C.c_uint64(0)
            cnt = L.ac_scan_count_range(ac, addr + w, b - w, a - w, n_owned - w, base + w, C.byref(chk))   # (ctypes drops the GIL)
            out[k] = (cnt, chk.value)

    th = [threading.Thread(target=work) for _ in range(min(threads, max(1, len(jobs))))]
    [t.start() for t in th]
    [t.join() for t in th]
    L.ac_free(ac)
    o.close()
    return sum(c for c, _ in out), sum(s for _, s in out) & (2**64 - 1)
