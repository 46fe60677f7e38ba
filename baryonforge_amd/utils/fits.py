"""
HEALPix maps in FITS files without healpy: what `hp.read_map(path)` does for the reference's
`LightconeShell(path=...)` (BaryonForge/utils/io.py:346-347), restated on numpy alone.

A HEALPix map file is a primary HDU without data followed by one BINTABLE extension (HEALPix "Facility Installation
Guidelines"; healpy.write_map writes columns of 1024 values per row): 2880-byte blocks of 80-character header cards up to
END, then the table as big-endian rows.  `read_healpix_map` parses exactly that -- column formats rL/B/I/J/K/E/D, TSCAL /
TZERO, implicit or explicit (partial-sky) indexing, BAD_DATA -> UNSEEN -- and returns the map in RING order, as read_map does
by default (a NESTED file is reordered with the integer ring -> (x, y, face) -> nest arithmetic of healpix_cxx).  When healpy
is importable it is used instead, so that installations which have it get its reader bit for bit.
"""
import re

import numpy as np

UNSEEN = -1.6375e30
_BLOCK = 2880
_TFORM = re.compile(r"^\s*(\d*)\s*([LXBIJKAED])")
_DTYPES = {"L": "u1", "B": "u1", "I": ">i2", "J": ">i4", "K": ">i8", "E": ">f4", "D": ">f8", "A": "S1"}


def _parse_value(txt):
    txt = txt.split("/")[0].strip() if not txt.lstrip().startswith("'") else txt
    t = txt.strip()
    if t.startswith("'"):
        end = t.find("'", 1)
        while end != -1 and t[end:end + 2] == "''":              # a doubled quote inside the string
            end = t.find("'", end + 2)
        return t[1:end if end != -1 else None].rstrip()
    if t in ("T", "F"):
        return t == "T"
    try:
        return int(t)
    except ValueError:
        try:
            return float(t.replace("D", "E"))
        except ValueError:
            return t


def _read_header(buf, pos):
    """cards of the header starting at byte `pos` -> (dict, position of the data)"""
    cards = {}
    while True:
        block = bytes(buf[pos:pos + _BLOCK])                        # (a memory map or bytes: headers are a few blocks)
        if len(block) < _BLOCK:
            raise ValueError("truncated FITS file (header)")
        pos += _BLOCK
        for i in range(0, _BLOCK, 80):
            card = block[i:i + 80].decode("ascii", errors="replace")
            key = card[:8].strip()
            if key == "END":
                return cards, pos
            if card[8:10] == "= " and key:
                cards.setdefault(key, _parse_value(card[10:]))


def _data_bytes(h):
    naxis = int(h.get("NAXIS", 0))
    if naxis == 0:
        return 0
    n = abs(int(h["BITPIX"])) // 8
    for k in range(1, naxis + 1):
        n *= int(h[f"NAXIS{k}"])
    n = (n + int(h.get("PCOUNT", 0))) * int(h.get("GCOUNT", 1))
    return n


def isqrt64(v):
    """floor(sqrt(v)) of non-negative int64 values, exactly"""
    v = np.asarray(v, dtype=np.int64)
    r = np.floor(np.sqrt(v.astype(np.float64))).astype(np.int64)
    r = np.where(r * r > v, r - 1, r)
    return np.where((r + 1) * (r + 1) <= v, r + 1, r)


def _spread_bits(v):
    v = np.asarray(v, dtype=np.uint64)
    v = (v | (v << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
    v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    v = (v | (v << np.uint64(2))) & np.uint64(0x3333333333333333)
    v = (v | (v << np.uint64(1))) & np.uint64(0x5555555555555555)
    return v


_JRLL = np.array([2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4], dtype=np.int64)
_JPLL = np.array([1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7], dtype=np.int64)


def ring2nest(nside, ipix):
    """NESTED index of RING pixels (nside a power of two): healpix_cxx's ring2xyf + xyf2nest, integers only"""
    nside = int(nside)
    if nside < 1 or nside & (nside - 1):
        raise ValueError("the NESTED scheme needs NSIDE to be a power of two")
    p = np.asarray(ipix, dtype=np.int64)
    npix, ncap, nl2 = 12 * nside * nside, 2 * nside * (nside - 1), 2 * nside
    north, south = p < ncap, p >= npix - ncap
    # north polar cap
    ir_n = (1 + isqrt64(1 + 2 * np.where(north, p, 0))) >> 1
    iphi_n = (p + 1) - 2 * ir_n * (ir_n - 1)
    face_n = (iphi_n - 1) // np.maximum(ir_n, 1)
    # equatorial belt
    ip = p - ncap
    tmp = ip // (4 * nside)
    ir_e = tmp + nside
    iphi_e = ip - tmp * 4 * nside + 1
    ks_e = (ir_e + nside) & 1
    ire = tmp + 1
    irm = nl2 + 1 - tmp
    ifm = (iphi_e - ire // 2 + nside - 1) // nside
    ifp = (iphi_e - irm // 2 + nside - 1) // nside
    face_e = np.where(ifp == ifm, ifp | 4, np.where(ifp < ifm, ifp, ifm + 8))
    # south polar cap
    ips = npix - p
    ir_s = (1 + isqrt64(np.maximum(2 * np.where(south, ips, 1) - 1, 0))) >> 1
    iphi_s = 4 * ir_s + 1 - (ips - 2 * ir_s * (ir_s - 1))
    face_s = 8 + (iphi_s - 1) // np.maximum(ir_s, 1)
    iring = np.where(north, ir_n, np.where(south, 2 * nl2 - ir_s, ir_e))
    iphi = np.where(north, iphi_n, np.where(south, iphi_s, iphi_e))
    nr = np.where(north, ir_n, np.where(south, ir_s, nside))
    kshift = np.where(north | south, 0, ks_e)
    face = np.clip(np.where(north, face_n, np.where(south, face_s, face_e)), 0, 11)
    irt = iring - _JRLL[face] * nside + 1
    ipt = 2 * iphi - _JPLL[face] * nr - kshift - 1
    ipt = np.where(ipt >= nl2, ipt - 8 * nside, ipt)
    ix = (ipt - irt) >> 1
    iy = (-ipt - irt) >> 1
    inter = (_spread_bits(ix) | (_spread_bits(iy) << np.uint64(1))).astype(np.int64)
    return face * nside * nside + inter


def _native(col):
    return np.ascontiguousarray(col.astype(col.dtype.newbyteorder("=")))


def read_healpix_map(path, field=0, hdu=1, nest=False, use_healpy=None):
    """The `field`-th map of HDU `hdu` of a HEALPix FITS file as a 1-D array in RING order (nest=True: NESTED), file dtype kept --
    `healpy.read_map(path, field=field, hdu=hdu, nest=nest)`.  use_healpy=None: healpy's own reader when it can be imported."""
    if use_healpy is None or use_healpy:
        try:
            import healpy as hp
            return hp.read_map(path, field=field, hdu=hdu, nest=nest)
        except ImportError:
            if use_healpy:
                raise
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    pos, k = 0, 0
    while True:
        hdr, data0 = _read_header(buf, pos)
        nbytes = _data_bytes(hdr)
        if k == hdu:
            break
        pos = data0 + (nbytes + _BLOCK - 1) // _BLOCK * _BLOCK
        k += 1
        if pos >= buf.size:
            raise ValueError(f"{path}: no HDU {hdu}")
    if str(hdr.get("XTENSION", "")).strip() != "BINTABLE":
        raise ValueError(f"{path}: HDU {hdu} is not a binary table")
    nrows, rowbytes, nfields = int(hdr["NAXIS2"]), int(hdr["NAXIS1"]), int(hdr["TFIELDS"])
    fields = []
    for i in range(1, nfields + 1):
        m = _TFORM.match(str(hdr[f"TFORM{i}"]))
        if not m or m.group(2) == "X":
            raise NotImplementedError(f"{path}: column format {hdr[f'TFORM{i}']!r}")
        rep = int(m.group(1)) if m.group(1) else 1
        fields.append((f"c{i}", _DTYPES[m.group(2)], (rep,)))
    row = np.dtype(fields)
    if row.itemsize != rowbytes:
        raise ValueError(f"{path}: NAXIS1 = {rowbytes} but the TFORMs add up to {row.itemsize} bytes")
    if data0 + nrows * rowbytes > buf.size:
        raise ValueError(f"{path}: truncated FITS file (table)")
    table = np.ndarray((nrows,), dtype=row, buffer=buf, offset=data0)
    explicit = str(hdr.get("INDXSCHM", "IMPLICIT")).strip().upper() == "EXPLICIT"
    col = field + 1 + (1 if explicit else 0)
    if col > nfields:
        raise ValueError(f"{path}: field {field} does not exist ({nfields - (1 if explicit else 0)} map column(s))")
    m = _native(table[f"c{col}"].reshape(-1))
    scal, zero = hdr.get(f"TSCAL{col}"), hdr.get(f"TZERO{col}")
    if (scal not in (None, 1, 1.0)) or (zero not in (None, 0, 0.0)):
        m = m * (1.0 if scal is None else float(scal)) + (0.0 if zero is None else float(zero))
    nside = hdr.get("NSIDE")
    if explicit:
        if nside is None:
            raise ValueError(f"{path}: a partial-sky map without NSIDE")
        pix = _native(table["c1"].reshape(-1)).astype(np.int64)
        full = np.full(12 * int(nside) ** 2, UNSEEN, dtype=m.dtype if m.dtype.kind == "f" else np.float64)
        full[pix] = m
        m = full
    if nside is not None and m.size != 12 * int(nside) ** 2:
        if m.size > 12 * int(nside) ** 2 and not np.any(m[12 * int(nside) ** 2:]):
            m = m[:12 * int(nside) ** 2]                               # (row padding of a writer that filled the last row)
        else:
            raise ValueError(f"{path}: {m.size} values for NSIDE = {nside}")
    bad = hdr.get("BAD_DATA")
    if bad is not None and m.dtype.kind == "f":
        m = np.where(m == bad, UNSEEN, m).astype(m.dtype)
    file_nested = str(hdr.get("ORDERING", "RING")).strip().upper().startswith("NEST")
    if file_nested != bool(nest):
        ns = int(round(np.sqrt(m.size / 12.0)))
        r2n = ring2nest(ns, np.arange(m.size, dtype=np.int64))
        if file_nested:                                                # NESTED file -> RING map
            m = m[r2n]
        else:                                                          # RING file -> NESTED map
            out = np.empty_like(m)
            out[r2n] = m
            m = out
    return np.ascontiguousarray(m)


def _card(key, value, comment=""):
    if isinstance(value, bool):
        v = f"{'T' if value else 'F':>20}"
    elif isinstance(value, (int, np.integer)):
        v = f"{int(value):>20d}"
    elif isinstance(value, float):
        v = f"{value:>20.13E}"
    else:
        v = "'" + f"{str(value):<8}" + "'"
        v = f"{v:<20}"
    return f"{key:<8}= {v} / {comment}"[:80].ljust(80)


def write_healpix_map(path, m, nest=False, column_rows=1024, extra_cards=()):
    """One full-sky map as a HEALPix FITS file of the layout healpy.write_map produces (one column of `column_rows` values per
    row; float32 / float64 kept) -- for round-trip tests and for handing maps to tools that read FITS."""
    m = np.asarray(m)
    if m.ndim != 1 or m.dtype.kind != "f" or m.dtype.itemsize not in (4, 8):
        raise ValueError("a 1-D float32 / float64 map")
    nside = int(round(np.sqrt(m.size / 12.0)))
    if 12 * nside * nside != m.size:
        raise ValueError("not a HEALPix map size")
    rep = column_rows if m.size % column_rows == 0 else 1
    code = "E" if m.dtype.itemsize == 4 else "D"
    primary = [_card("SIMPLE", True, "conforms to FITS standard"), _card("BITPIX", 8), _card("NAXIS", 0), _card("EXTEND", True)]
    ext = [_card("XTENSION", "BINTABLE", "binary table extension"), _card("BITPIX", 8), _card("NAXIS", 2),
           _card("NAXIS1", rep * m.dtype.itemsize, "width of table in bytes"), _card("NAXIS2", m.size // rep, "number of rows"),
           _card("PCOUNT", 0), _card("GCOUNT", 1), _card("TFIELDS", 1), _card("TTYPE1", "TEMPERATURE"),
           _card("TFORM1", f"{rep}{code}"), _card("PIXTYPE", "HEALPIX", "HEALPIX pixelisation"),
           _card("ORDERING", "NESTED" if nest else "RING", "Pixel ordering scheme"), _card("NSIDE", nside),
           _card("FIRSTPIX", 0), _card("LASTPIX", m.size - 1), _card("INDXSCHM", "IMPLICIT"), _card("OBJECT", "FULLSKY")]
    ext += [_card(k, v) for k, v in extra_cards]

    def block(cards):
        txt = "".join(cards) + "END".ljust(80)
        return (txt + " " * (-len(txt) % _BLOCK)).encode("ascii")
    data = m.astype(m.dtype.newbyteorder(">")).tobytes()
    with open(path, "wb") as f:
        f.write(block(primary))
        f.write(block(ext))
        f.write(data)
        f.write(b"\0" * (-len(data) % _BLOCK))
