"""Dictionary streams (zstd/encoder.go:257-428 after Encoder.Reset with a dictionary): the oracle's restatement of
both nextBlock forms, on the CPU."""
import pytest

import corpora
import test_oracle_kats as tk


@pytest.mark.parametrize("level", [1, 2, 3])
def test_oracle_dictionary_streams_roundtrip_and_forms(oracle, level):
    """Frames of both forms decode back with the dictionary (libzstd for the full-format dictionaries, the in-repo decoder for
    the raw-content one, whose frames carry an id libzstd does not accept); a raw-content dictionary has no literal table, so the
    two forms agree; with a literal table huff0 keeps (the skewed one) they do not, and the asynchronous form — the one that
    sees the table — is never the larger."""
    t = corpora.corpus("T", 4, 131072, first_unit=21).tobytes()
    raw = corpora.corpus("T", 1, 65536, seed=0x5EED0005).tobytes()
    blob, ins = tk._dict_fixture(oracle)
    sk, probs = tk.skewed_dict(blob)
    sku = tk.skewed_units(probs, sizes=(40, 300, 1000, 4000, 70000), seeds=2)
    bs = 65536 if level == 1 else 131072
    for name, okw, dct, extra in (("raw", dict(dict_id=7, dict_content=raw), raw, [raw[100:40000]]),
                                  ("d0", dict(dict_blob=blob), blob, [ins[1], ins[2] + ins[4]]),
                                  ("skewed", dict(dict_blob=sk), sk, sku)):
        a = oracle.ZstdOracle(level=level, **okw)
        s = oracle.ZstdOracle(level=level, concurrent=1, **okw)
        differ = 0
        for u in extra + [t[:bs], t[:bs + 1], t[:3 * bs + 77], t[:1000], b""]:
            for cuts in ((), (len(u) // 3,), (len(u),)):
                fa, fs = a.encode_stream(u, cuts), s.encode_stream(u, cuts)
                for fr in (fa, fs):
                    if not u and not fr:
                        continue
                    dec = oracle.zstd_decode(fr, len(u) + 16, dict_content=raw) if name == "raw" else oracle.zstd_decompress(fr, len(u) + 16, dict_content=dct)
                    assert dec == u, (name, len(u), cuts)
                if u:  # the frame names its dictionary (frameenc.go:60-80)
                    assert fa[4] & 3 != 0
                differ += fa != fs
                assert len(fa) <= len(fs), (name, len(u), cuts)
                # below one block with no Flush before Close, both are the EncodeAll frame (encoder.go:272-288)
                if 0 < len(u) < bs and cuts == ():
                    assert fa == fs == a.encode_all(u)
        assert (differ > 0) == (name == "skewed"), (name, differ)
