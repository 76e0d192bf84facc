"""Crafted FASTQ inputs that sit on the decision boundaries of the GPU kernels (path switches, exception paths).
Shared by the CPU tests (the oracle accepts them and round-trips them) and the GPU parity tests."""
import random


def _fq(recs):
    return b"".join(b"@%s\n%s\n+\n%s\n" % r for r in recs)


def build():
    rnd = random.Random(7)
    cases = {}
    # title lines around 64 and 128 bytes: the word-parallel title kernels take titles of up to 64 / 128 bytes
    for tl in (63, 64, 65, 66, 127, 128, 129, 130):      # 64 / 128: one / two mask words, beyond: the byte-serial kernels
        recs = []
        for i in range(4000):
            t = b"r%d " % (i * 7919 % 100000)
            t = t + b"x" * (tl - 2 - len(t))        # '@' + t + '\n' = tl bytes
            recs.append((t, b"ACGT" * 9, bytes(rnd.choice(b"ABCDEFGHI") for _ in range(36))))
        cases["title_len_%d" % tl] = (_fq(recs), 2)
    # empty reads and reads of 1..50 bases: tails of every length, records without a single 16-byte chunk
    recs = []
    for i in range(6000):
        L = 0 if i % 97 == 5 else 1 + (i * 31 % 50)
        recs.append((b"e.%d" % i, bytes(rnd.choice(b"ACGT") for _ in range(L)), bytes(rnd.choice(b"!#5?IJ") for _ in range(L))))
    cases["len_0_to_50"] = (_fq(recs), 3)
    # 0..7 transferred N per record (the position list holds four) and all-N reads
    recs = []
    for i in range(5000):
        L = 40
        s = bytearray(rnd.choice(b"ACGT") for _ in range(L))
        q = bytearray(rnd.choice(b"5?IJ") for _ in range(L))
        nn = L if i % 50 == 0 else i % 8
        for p in rnd.sample(range(L), nn):
            s[p] = ord("N")
            q[p] = ord(rnd.choice("!#"))
        recs.append((b"n.%d" % i, bytes(s), bytes(q)))
    cases["n_counts"] = (_fq(recs), 2)
    # quality bytes outside the 64 LDS columns ('a'..'~') and IUPAC codes with high quality (Huffman-coded DNA)
    recs = []
    for i in range(5000):
        L = 36
        s = bytearray(rnd.choice(b"ACGT") for _ in range(L))
        q = bytearray(rnd.choice(b"IJKabcxyz~") for _ in range(L))
        if i % 3 == 0:
            s[i % L] = ord(rnd.choice("RYKMSWN"))
        recs.append((b"q.%d" % i, bytes(s), bytes(q)))
    cases["odd_quals_iupac"] = (_fq(recs), 2)
    # records just below and above 2 KB (register-streaming vs wave-per-record kernels)
    for L in (980, 1015, 1030):
        recs = [(b"L.%d" % i, bytes(rnd.choice(b"ACGT") for _ in range(L)), bytes(rnd.choice(b"5?IJ") for _ in range(L)))
                for i in range(600)]
        cases["reclen_%d" % L] = (_fq(recs), 2)
    # lowercase bases: not ambiguity codes, never transferred, part of the DNA alphabet
    recs = [(b"lc.%d" % i, bytes(rnd.choice(b"ACGTacgtn") for _ in range(36)), bytes(rnd.choice(b"!#5I") for _ in range(36)))
            for i in range(5000)]
    cases["lowercase"] = (_fq(recs), 2)
    # the file's last line without a newline (the reference works from positions and still finds the record)
    recs = [(b"nn.%d" % i, bytes(rnd.choice(b"ACGT") for _ in range(36)), bytes(rnd.choice(b"5?IJ") for _ in range(36)))
            for i in range(3000)]
    cases["no_final_newline"] = (_fq(recs)[:-1], 3)
    # ---- found by the randomised runs (tools/fuzz_parity.py) ----
    # the file ends with "+\n" and an empty, unterminated quality line
    recs = [(b"el.%d" % i, bytes(rnd.choice(b"ACGT") for _ in range(i % 5)), bytes(rnd.choice(b"5?IJ") for _ in range(i % 5)))
            for i in range(3001)]
    recs[-1] = (b"el.3000", b"", b"")
    cases["empty_last_quality_no_newline"] = (_fq(recs)[:-1], 3)
    # constant numeric fields beyond int32 / uint32 (stored once as text; the reference never uses the wrapped value)
    recs = [(b"c%d 4294967295 x:%d 99999999999" % (i % 7, i), bytes(rnd.choice(b"ACGT") for _ in range(36)),
             bytes(rnd.choice(b"5?IJ") for _ in range(36))) for i in range(5000)]
    cases["const_numeric_beyond_int32"] = (_fq(recs), 2)
    # twelve varying numeric fields (more than the eight the first engine versions tracked)
    recs = [(b"M" + b":".join(b"%d" % ((i * (k + 3) + k * k) % (10 ** (1 + k % 6))) for k in range(12)),
             bytes(rnd.choice(b"ACGT") for _ in range(36)), bytes(rnd.choice(b"5?IJ") for _ in range(36))) for i in range(6000)]
    cases["twelve_numeric_fields"] = (_fq(recs), 2)
    # eight varying text fields in each of ten sub-blocks: more text-field tables than the engine reserves up front
    def word(k):
        return bytes(rnd.choice(b"abcdefgh") for _ in range(1 + (k + rnd.randrange(3)) % 4))
    recs = [(b"T" + b" ".join(word(k) for k in range(8)), bytes(rnd.choice(b"ACGT") for _ in range(50)),
             bytes(rnd.choice(b"5?IJ") for _ in range(50))) for i in range(6000)]
    cases["many_text_fields"] = (_fq(recs), 10)
    # 3000-base reads: records of 6 KB, far longer than the reference's 500-byte overlap (the bytes behind a window
    # are the file's own)
    recs = [(b"long.%d" % i, bytes(rnd.choice(b"ACGTN") for _ in range(3000 + i % 11)),
             bytes(rnd.choice(b"#5?IJ") for _ in range(3000 + i % 11))) for i in range(300)]
    cases["reads_3000bp"] = (_fq(recs), 2)
    # titles between 65 and 128 bytes with a field longer than 64 bytes, digits across the word boundary of the masks,
    # and numeric fields that start in the second mask word
    recs = []
    for i in range(4000):
        t = b"X" * (60 + i % 9) + b" " + b"%d" % (10 ** 8 + i * 7919) + b":" + b"y" * (3 + i % 3) + b"_" + b"%d" % (i % 1000)
        recs.append((t, b"ACGT" * 9, bytes(rnd.choice(b"ABCDEFGHI") for _ in range(36))))
    cases["title_two_mask_words"] = (_fq(recs), 2)
    recs = []
    for i in range(3000):
        t = b"run" + b"Z" * (70 + i % 5) + b"/" + b"%d" % i        # one text field longer than 64 bytes
        recs.append((t, b"ACGT" * 9, bytes(rnd.choice(b"ABCDEFGHI") for _ in range(36))))
    cases["title_field_longer_than_64"] = (_fq(recs), 3)
    # 40 and 64 title fields (the field masks of the title kernels are 64 bits wide; the first versions stopped at 32):
    # numeric fields that vary, stay constant and switch between the two by group, with text fields between them -
    # titles below 128 bytes (word-parallel kernels, two mask words) and above (byte-serial kernels)
    for nfields, width in ((40, 1), (64, 1), (48, 3)):
        recs = []
        for i in range(5000):
            parts = []
            for k in range(nfields):
                if k % 7 == 3: parts.append(b"t" if (i // 64 + k) % 3 else b"u")                      # text, constant over stretches
                elif k % 5 == 0: parts.append(b"%d" % ((i * (k + 1)) % 10 ** width))                   # varying
                elif k % 5 == 1: parts.append(b"%d" % ((i // 32) % 10 ** width))                       # constant inside a group
                else: parts.append(b"%d" % (k % 10 ** width))                                          # constant
            recs.append((b":".join(parts), bytes(rnd.choice(b"ACGT") for _ in range(36)), bytes(rnd.choice(b"5?IJ") for _ in range(36))))
        cases["title_fields_%d_w%d" % (nfields, width)] = (_fq(recs), 3)
    # the smallest records there are: '@' alone as the title and an empty read (6 bytes), next to one-base reads
    recs = [(b"", b"", b"") if i % 3 else (b"", b"A", b"I") for i in range(9000)]
    cases["six_byte_records"] = (_fq(recs), 2)
    return cases
