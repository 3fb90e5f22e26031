#!/usr/bin/env python3
"""Copies the reference's own encoder-test INPUTS (fixtures, not sources) into tests/golden/ref_inputs/.

The GPU box has no /root/reference, so the parity tests that run the HIP path on the reference's inputs
(tests/test_gpu_ref_inputs.py) read these committed copies.  Re-run in the build container to refresh:

    python tests/golden/fetch_ref_inputs.py

Sources (all under /root/reference):
  zstd/testdata/fuzz/encode-corpus-raw.zip   zstd/fuzz_test.go:154 FuzzEncoding seed corpus
  zstd/testdata/comp-crashers.zip            zstd/encoder_test.go:68 TestEncoderRegression
  s2/testdata/enc_regressions.zip            s2/encode_test.go TestEncoderRegression
  testdata/{e.txt,gettysburg.txt,Mark.Twain-Tom.Sawyer.txt,sharnd.out,pi.txt,html.txt,pngdata.bin}
  zstd/testdata/z000028                      the input of z000028.zst
"""
import os
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
DST = os.path.join(HERE, "ref_inputs")
FILES = [
    "zstd/testdata/fuzz/encode-corpus-raw.zip",
    "zstd/testdata/comp-crashers.zip",
    "s2/testdata/enc_regressions.zip",
    "testdata/e.txt", "testdata/gettysburg.txt", "testdata/Mark.Twain-Tom.Sawyer.txt", "testdata/sharnd.out",
    "testdata/pi.txt", "testdata/html.txt", "testdata/pngdata.bin",
    "zstd/testdata/z000028",
]

if __name__ == "__main__":
    os.makedirs(DST, exist_ok=True)
    for f in FILES:
        shutil.copyfile(os.path.join(REF, f), os.path.join(DST, os.path.basename(f)))
        print(f, os.path.getsize(os.path.join(DST, os.path.basename(f))))
