"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's FASTA/Q record reader.

Follows seqio/fastx/reader.go line by line: Reader.Read (:233-369, the window loop with `needMoreCheckOfBuf`, the
format check :272-306, the delimiter search :310-366) and parseRecord (:372-471, without ID parsing and validation).
Nothing under bio_amd/ or include/ may import this; tests compare the product's reader (bio_amd/csrc/fastx.cpp)
against it on the reference's own test files and on synthetic edge cases.
"""


class NotFastx(Exception):      # ErrNotFASTXFormat reader.go:16
    pass


class BadFastq(Exception):      # ErrBadFASTQFormat :19
    pass


class UnequalSeqQual(Exception):  # ErrUnequalSeqAndQual :22
    pass


def _drop_cr(b: bytes) -> bytes:  # util.go dropCR
    return b[:-1] if b.endswith(b"\r") else b


def _drop_lf(b: bytes) -> bytes:
    return b[:-1] if b.endswith(b"\n") else b


def read_records(data: bytes, bufsize: int = 1 << 30):
    """-> (records [(head, seq, qual-or-None)], is_fastq or None, error or None)"""
    records = []
    pos = 0                      # file position
    buf = b""
    r = 0
    last_part = finished = False
    need_more = False
    check_type = True
    is_fastq = None
    delim = None
    buffer = bytearray()

    def parse():  # parseRecord :372 -> (shorter_qual, err, record)
        p = bytes(buffer)
        j = p.find(b"\n")
        seq = bytearray()
        qual = bytearray()
        if j > 0:
            head = _drop_cr(p[0:j])
            rr = j + 1
            if not is_fastq:
                while True:
                    k = p.find(b"\n", rr)
                    if k >= 0:
                        seq += _drop_cr(p[rr:k])
                        rr = k + 1
                        continue
                    seq += _drop_cr(p[rr:])
                    break
            else:
                is_qual = False
                while True:
                    k = p.find(b"\n", rr)
                    if k >= 0:
                        if k - rr > 0 and p[rr:rr + 1] == b"+" and not is_qual:
                            is_qual = True
                        elif is_qual:
                            qual += _drop_cr(p[rr:k])
                        else:
                            seq += _drop_cr(p[rr:k])
                        rr = k + 1
                        continue
                    if is_qual:
                        qual += _drop_cr(p[rr:])
                    break
                if len(seq) != len(qual):
                    return len(seq) > len(qual), UnequalSeqQual(), None
        else:
            head = _drop_cr(_drop_lf(p))
        if len(head) == 0 and len(seq) == 0:
            return False, EOFError(), None
        return False, None, (bytes(head), bytes(seq), bytes(qual) if is_fastq else None)

    while True:  # successive Read() calls
        if last_part and finished:
            return records, is_fastq, None
        got = None
        while got is None:
            if not need_more and not last_part:
                chunk = data[pos:pos + bufsize]
                pos += len(chunk)
                if len(chunk) == 0 or pos >= len(data):   # fh.Read returning io.EOF (together with the last bytes or alone)
                    last_part = True
                buf = chunk
                r = 0
            if check_type:
                pn = 0
                for i, c in enumerate(buf):
                    if c == 0x3E:
                        check_type, is_fastq, delim, r = False, False, b">", i + 1
                        break
                    if c == 0x40:
                        check_type, is_fastq, delim, r = False, True, b"@", i + 1
                        break
                    if c == 0x0A:
                        pn += 1
                        if pn > 100 and i > 10240:
                            return records, is_fastq, NotFastx()
                    else:
                        return records, is_fastq, NotFastx()
                check_type = False
                if delim is None:          # only newlines (or nothing): no delimiter will ever be found
                    return records, is_fastq, None
            while True:  # FORSEARCH
                i = buf.find(delim, r)
                if i >= 0:
                    i -= r
                    if i > 0:
                        last_byte = buf[r + i - 1]
                    else:
                        last_byte = buffer[-1] if len(buffer) else 0
                    if last_byte == 0x0A:
                        if i > 0:
                            buffer += _drop_cr(buf[r:r + i - 1])
                        else:
                            buffer += b"\n"
                        shorter, err, rec = parse()
                        if is_fastq and isinstance(err, UnequalSeqQual):
                            if shorter:
                                buffer += b"\n" + delim
                                need_more = True
                                r += i + 1
                                continue
                            return records, is_fastq, BadFastq()
                        buffer.clear()
                        need_more = True
                        r += i + 1
                        if isinstance(err, EOFError):
                            return records, is_fastq, None
                        got = rec
                        break
                    buffer += buf[r:r + i + 1]
                    r += i + 1
                    need_more = True
                    continue
                buffer += buf[r:]
                if last_part:
                    _, err, rec = parse()
                    if isinstance(err, EOFError):
                        return records, is_fastq, None
                    if err is not None:
                        return records, is_fastq, err
                    buffer.clear()
                    finished = True
                    got = rec
                    break
                need_more = False
                break
        records.append(got)


def bgzf_compress(data: bytes, block: int = 0xff00, level: int = 6, rng=None) -> bytes:
    """BGZF (SAM specification 4.1): gzip members of at most 64 KiB, each with the 'BC' extra subfield = its own size - 1, and the
    empty end-of-file member.  rng: random block sizes (block boundaries everywhere)."""
    import struct
    import zlib
    out = bytearray()
    i = 0
    chunks = []
    while i < len(data):
        n = block if rng is None else rng.randint(1, block)
        chunks.append(data[i:i + n])
        i += n
    chunks.append(b"")
    for c in chunks:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(c) + co.flush()
        bsize = 12 + 6 + len(comp) + 8
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += comp + struct.pack("<II", zlib.crc32(c) & 0xffffffff, len(c))
    return bytes(out)
