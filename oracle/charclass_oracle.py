"""TEST INFRASTRUCTURE (the checker, never the product): brute-force matcher for character-class pattern files.

Parity: UNPINNED against the reference.  mickeyjoe666/PHFPFAC sketches this front end in
regex_GPU_PHF/CreateTable/charset_table_reorder.c (grammar: fgetc_set :131-168 on top of fgetc_ext ctdef.h:37-99,
chain NFA :45-129, subset construction :321-427), but that file is included by nothing, refers to undefined globals
(PFAC_table, num_output, outputs :483,504-505) and cannot be compiled, and the repository holds no fixture for it.  This
module restates the GRAMMAR from that source and defines the obvious PFAC semantics on top of it:

    for every start offset i, every pattern p (a fixed-length sequence of byte sets) with input[i+j] in p[j] for all j
    and i + len(p) <= N is reported as (i, id); order: position, then pattern length, then pattern id (1-based line).

It shares no code with phfpfac_amd/csrc/pfac_table.c: own escape reader, own class parser, and a matcher that never
builds an automaton (one vectorised AND per pattern element).
"""
import numpy as np

EOL = 0x10A


class _Cur:
    def __init__(self, b):
        self.b, self.i = b, 0

    def getc(self):
        if self.i < len(self.b):
            self.i += 1
            return self.b[self.i - 1]
        return -1

    def unget(self, ch):
        if ch != -1:
            self.i -= 1


def _scan_number(c, base, width):
    """glibc fscanf("%<width>o"/"%<width>x"): returns the value, or None when nothing was matched."""
    ch = c.getc()
    while ch in (0x20, 9, 10, 11, 12, 13):
        ch = c.getc()
    if ch == -1:
        return None
    neg, digits, acc = False, 0, 0
    if ch in (0x2D, 0x2B):
        neg = ch == 0x2D
        width -= 1
        ch = c.getc()
    if width != 0 and ch == 0x30:
        width -= 1
        digits = 1
        ch = c.getc()
        if width != 0 and ch in (0x78, 0x58) and base == 16:
            width -= 1
            ch = c.getc()
    while ch != -1 and width != 0:
        s = chr(ch)
        d = int(s, 16) if s in "0123456789abcdefABCDEF" else 99
        if d >= base:
            break
        acc = acc * base + d
        digits += 1
        width -= 1
        ch = c.getc()
    c.unget(ch)
    if digits == 0:
        return None
    return (-acc if neg else acc) & 0xFFFFFFFF


def _getc_ext(c):
    c0 = c.getc()
    if c0 == 0x5C:
        c1 = c.getc()
        if c1 == -1:
            return c0
        if 0x30 <= c1 <= 0x39:
            c.unget(c1)
            v = _scan_number(c, 8, 3)
            return (v or 0) & 0xFF
        simple = {ord("a"): 7, ord("b"): 8, ord("t"): 9, ord("n"): 10, ord("v"): 11, ord("f"): 12, ord("r"): 13,
                  ord("'"): ord("'"), ord('"'): ord('"'), 0x5C: 0x5C}
        if c1 in simple:
            return simple[c1]
        if c1 == ord("x"):
            v = _scan_number(c, 16, 2)
            return (v or 0) & 0xFF
        c.unget(c1)
        return c0
    if c0 == 10:
        return EOL
    return c0


def parse(image: bytes):
    """-> list of patterns, each a list of 256-entry boolean arrays (one per element)."""
    c = _Cur(image)
    pats = []
    while c.i < len(image):
        elems = []
        while True:
            at_end = c.i >= len(image)
            ch = _getc_ext(c)
            if ch == EOL:
                break
            if ch == -1 and at_end:
                raise ValueError("pattern file must end with a newline")
            s = np.zeros(256, dtype=bool)
            if ch == ord("["):
                setting, have_l, l = True, False, 0
                ch = _getc_ext(c)
                if ch == ord("^"):
                    s[:] = True
                    setting = False
                    ch = _getc_ext(c)
                while ch != ord("]"):
                    if ch == EOL or ch == -1:
                        raise ValueError("class not closed")
                    if ch == ord("-") and have_l:
                        r = _getc_ext(c)
                        if r == EOL or r == -1:
                            raise ValueError("class not closed")
                        s[l: (r & 0xFF) + 1] = setting
                    else:
                        l = ch & 0xFF
                        s[l] = setting
                        have_l = True
                    ch = _getc_ext(c)
            else:
                s[ch & 0xFF] = True
            elems.append(s)
        if not elems:
            raise ValueError("empty pattern")
        pats.append(elems)
    return pats


def match(image: bytes, data: np.ndarray):
    """-> (pos int64[], id int32[]) ordered by (position, pattern length, pattern id)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    n = data.size
    pos_l, id_l, len_l = [], [], []
    for pid, elems in enumerate(parse(image), start=1):
        L = len(elems)
        if L > n:
            continue
        ok = np.ones(n - L + 1, dtype=bool)
        for j, s in enumerate(elems):
            ok &= s[data[j: n - L + 1 + j]]
        p = np.flatnonzero(ok)
        pos_l.append(p)
        id_l.append(np.full(p.size, pid, dtype=np.int32))
        len_l.append(np.full(p.size, L, dtype=np.int32))
    if not pos_l:
        return np.empty(0, np.int64), np.empty(0, np.int32)
    pos, ids, lens = np.concatenate(pos_l), np.concatenate(id_l), np.concatenate(len_l)
    order = np.lexsort((ids, lens, pos))
    return pos[order].astype(np.int64), ids[order]
