#!/bin/bash
# VGPR / scratch / LDS of every kernel in a built libdifusion.so (from the code object's metadata notes)
so=${1:-di_fusion_amd/libdifusion.so}
L=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
$L/clang-offload-bundler --type=o --unbundle --input=$so --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/co.o 2>/dev/null || \
  python3 - "$so" "$tmp/co.o" <<'PY'
import sys
d=open(sys.argv[1],'rb').read()
i=d.find(b'\x7fELF', d.find(b'__CLANG_OFFLOAD_BUNDLE__'))
open(sys.argv[2],'wb').write(d[i:])
PY
$L/llvm-readelf --notes $tmp/co.o | python3 -c "
import sys,re
txt=sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+).*?(?=\n\s+- \.agpr_count|\Z)', txt, re.S):
    pass
blocks=txt.split('- .agpr_count')
for b in blocks[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)', b) or [None,'?'])[1]
    name=g('name')
    print(f\"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} agpr {b.split()[0].strip(':'):>3s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} spill {g('vgpr_spill_count'):>3s} lds {g('group_segment_fixed_size'):>6s}\")
"
rm -rf $tmp
