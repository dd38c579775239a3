#!/usr/bin/env python3
"""Emit the marching-cubes case tables as C headers.

The 256-case triangle table is the classic public-domain table from Paul Bourke, "Polygonising a scalar field"
(1994, after Lorensen & Cline 1987, table by Cory Gene Bloyd) — the same public table every marching-cubes
implementation (including the reference's `ext/marching_cubes/mc_data.cuh:40,54`) carries.  It is written here
in a compact row form (one string of edge ids per case, hex digits); the EDGE table is not stored at all — it
is derived from the cube topology (edge e is cut iff its two end corners have different sign bits).

Cube convention (the reference's, `mc_interp_kernel.cu:240-295`):
  corners 0 (0,0,0) 1 (1,0,0) 2 (1,1,0) 3 (0,1,0) 4 (0,0,1) 5 (1,0,1) 6 (1,1,1) 7 (0,1,1)
  edges   0-1 1-2 2-3 3-0 4-5 5-6 6-7 7-4 0-4 1-5 2-6 3-7

Outputs (identical content, two consumers that must not depend on each other):
  di_fusion_amd/csrc/mc_tables.inc   (product, included by the HIP kernel)
  oracle/mc_tables_oracle.inc        (oracle, included by oracle/mc_oracle.c)

`--check-reference` parses the reference header (build container only) and asserts equality; the SHA-256 of the
table bytes is pinned in tests/test_oracle_golden.py so the check also runs where the reference is absent.
"""
import hashlib
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent

EDGE_CORNERS = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]

TRI_ROWS = """
-
083
019
183981
12a
08312a
92a029
2832a8a98
3b2
0b28b0
19023b
1b219b98b
3a1ba3
0a108a8ba
3903b9ba9
98aa8b
478
430734
019847
419471731
12a847
34730412a
92a902847
2a9297273794
8473b2
b47b24204
90184723b
47b94b9b2921
3a13ba784
1ba14b1047b4
47890b9bab03
47b4b99ba
954
954083
054150
854835315
12a954
30812a495
52a542402
2a5325354348
95423b
0b208b495
05401523b
21525828b485
a3ba13954
4950818a18ba
54050b5bab03
54858aa8b
978579
930953573
078017157
153357
978957a12
a12950530573
802825857a52
2a5253357
7957893b2
95797292027b
23b018178157
b21b17715
958857a13a3b
5705097b010aba0
ba0b03a50807570
ba57b5
a65
0835a6
9015a6
1831985a6
165261
165126308
965906026
598582526328
23ba65
b08b20a65
01923b5a6
5a61929b298b
63b653513
08b0b50515b6
3b6036065059
65969bb98
5a6478
4304736 5a
1905a6847
a6519717379 4
612651478
125526304347
847905065026
739794329596269
3b2784a65
5a647242027b
01947823b5a6
9219b294b7b45a6
8473b53515b6
51b5b610b7b404b
059065036b63847
6596 9b4797b9
a4964a
4a649a083
a01a60640
83181686461a
149124264
308129249264
024426
832824426
a49a64b23
08228b49a4a6
3b2016064 61a
64161a48121b8b1
964936913b63
8b1810b61914641
3b6360064
648b68
7a678a89a
0730a709a67a
a671a7178180
a67a71173
126168189867
269291679093739
780706602
732672
23ba68a89867
2072 7b09767a9a7
1801781a767a23b
b21b17a61671
896867916b63136
091b67
7807063b0b60
7b6
76b
308b76
019b76
819831b76
a126b7
12a3086b7
2902a96b7
6b72a3a83a98
723627
708760620
276237019
162186198876
a76a17137
a7617a187108
03707a0a96a7
76a7a88a9
684b86
36b306046
86b846901
946963931b36
6846b82a1
12a30b06b046
4b846b0292a9
a93a32943b36463
823842462
042462
190234246438
194142246
8138618466a1
a10a06604
4634386a3039a93
a946a4
49576b
083495b76
501540 76b
b76834354315
954a1276b
6b712a083495
76b54a42a402
348354325a52b76
723762549
954086062687
362376150540
628687218485158
954a16176137
16a176107870954
40a4a503a6a737a
76a7a854a48a
695 6b9b89
36b063056095
0b805b01556b
6b3635531
12a95b9b8b56
0b306b0965691 2a
b85b56805a52025
6b36352a3a53
58952856 2382
956960062
158180568382628
156216
1361 6a386569896
a10a06950560
0385 6a
a56
b5a75b
b5ab75830
5b75ab190
a75ab7981831
b12b71751
08312717572b
975927902 2b7
75272b592328982
25a235375
820852875a25
9015a35373a2
982921872a25752
135375
087071175
903935537
987597
5845a8ab8
5045b05abb30
01984a8aba45
ab4a45b34941314
2512852b8458
04b0b345b2b151b
0250592b5458b85
9452b3
25a352345384
5a2524420
3a235a385458019
5a2524192942
845853351
045105
845853905035
945
4b749b9ab
0834979b79ab
1ab1b414074b
314348 1a474ba b4
4b79b492b912
97 49b791b2b1083
b74b42240
b74b42834324
29a279237749
9a7974a27870207
37a3a274a1a040a
1a2874
491417713
491417081871
403743
487
9a8ab8
3093 9bb9a
01a0a88ab
31ab3a
12b1b99b8
3093 9b1292b9
02b80b
32b
2382 8aa89
9a2092
238 28a0181a8
1a2
138918
091
038
-
"""


def tri_table() -> np.ndarray:
    rows = [r.replace(" ", "") for r in TRI_ROWS.strip().splitlines()]
    assert len(rows) == 256, len(rows)
    t = -np.ones((256, 16), dtype=np.int32)
    for c, r in enumerate(rows):
        if r == "-":
            continue
        assert len(r) % 3 == 0 and len(r) <= 15, (c, r)
        t[c, :len(r)] = [int(ch, 16) for ch in r]
    return t


def edge_table() -> np.ndarray:
    e = np.zeros(256, dtype=np.int32)
    for c in range(256):
        m = 0
        for k, (a, b) in enumerate(EDGE_CORNERS):
            if ((c >> a) & 1) != ((c >> b) & 1):
                m |= 1 << k
        e[c] = m
    return e


def structural_checks(tri: np.ndarray, edge: np.ndarray):
    n_tri = 0
    for c in range(256):
        row = tri[c]
        used = row[row >= 0]
        assert len(used) % 3 == 0
        assert (row[len(used):] == -1).all()
        n_tri += len(used) // 3
        mask = 0
        for k in used:
            mask |= 1 << int(k)
        assert mask == edge[c], (c, bin(mask), bin(edge[c]))      # triangles use exactly the cut edges
    assert n_tri == 820, n_tri


def table_sha(tri: np.ndarray, edge: np.ndarray) -> str:
    return hashlib.sha256(edge.astype("<i4").tobytes() + tri.astype("<i4").tobytes()).hexdigest()


def emit(path: Path, tri, edge, prefix: str, qual: str):
    lines = ["// GENERATED by tools/gen_mc_tables.py — do not edit.  Classic public-domain marching-cubes case tables",
             "// (Bourke 1994 / Lorensen-Cline 1987); edge table derived from cube topology.",
             f"{qual} int {prefix}edge_table[256] = {{"]
    for i in range(0, 256, 16):
        lines.append("  " + ", ".join(f"0x{v:03x}" for v in edge[i:i + 16]) + ",")
    lines.append("};")
    lines.append(f"{qual} signed char {prefix}tri_table[256][16] = {{")
    for c in range(256):
        lines.append("  {" + ",".join(f"{v:2d}" for v in tri[c]) + "},")
    lines.append("};")
    lines.append(f"{qual} unsigned char {prefix}tri_count[256] = {{")
    cnt = [(tri[c] >= 0).sum() // 3 for c in range(256)]
    for i in range(0, 256, 32):
        lines.append("  " + ",".join(str(v) for v in cnt[i:i + 32]) + ",")
    lines.append("};")
    path.write_text("\n".join(lines) + "\n")


def parse_reference():
    src = Path("/root/reference/pytorch/system/ext/marching_cubes/mc_data.cuh").read_text()
    m = re.search(r"edgeTable\[256\]\s*=\s*\{(.*?)\};", src, re.S)
    e = np.array([int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", m.group(1))], dtype=np.int32)
    m = re.search(r"triangleTable\[256\]\[16\]\s*=\s*\{(.*?)\};", src, re.S)
    t = np.array([int(x) for x in re.findall(r"-?\d+", m.group(1))], dtype=np.int32).reshape(256, 16)
    return t, e


def main():
    tri, edge = tri_table(), edge_table()
    if "--check-reference" in sys.argv:
        rt, re_ = parse_reference()
        bad = [c for c in range(256) if not (rt[c] == tri[c]).all()]
        print("edge table equal:", bool((re_ == edge).all()), " tri rows differing:", bad)
        for c in bad[:300]:
            print(c, "mine", tri[c][tri[c] >= 0].tolist())
        if bad or not (re_ == edge).all():
            sys.exit(1)
    structural_checks(tri, edge)
    emit(ROOT / "di_fusion_amd/csrc/mc_tables.inc", tri, edge, "k_mc_", "static const")
    emit(ROOT / "oracle/mc_tables_oracle.inc", tri, edge, "mc_oracle_", "static const")
    print("sha256", table_sha(tri, edge))


if __name__ == "__main__":
    main()
