#!/usr/bin/env python
"""Writes benchdata/val_phonemes_32.txt: column 2 (the phonemised text) of the first 32 lines of the reference's
Data/val_list.txt (`wav|phonemes|speaker`, the LJSpeech validation list the reference ships as a data file).  Run in the build
container, where /root/reference exists; the GPU box only ever reads the committed fixture.  Data, not code: these are the
reference's own validation inputs for `TextCleaner` (text_utils.py:3-26) -- real, ragged phoneme strings (58-170 symbols)
instead of the bench's uniform 100-token rows (SURVEY.md section 8(d) "optional realistic phoneme strings")."""
import os
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/Data/val_list.txt"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "val_phonemes_32.txt")
rows = [line.rstrip("\n").split("|") for line in open(SRC, encoding="utf-8") if line.strip()][:32]
assert len(rows) == 32 and all(len(r) == 3 for r in rows)
with open(OUT, "w", encoding="utf-8") as f:
    for r in rows:
        f.write(r[1] + "\n")
print("%d utterances, %d..%d symbols -> %s" % (len(rows), min(len(r[1]) for r in rows), max(len(r[1]) for r in rows), OUT))
