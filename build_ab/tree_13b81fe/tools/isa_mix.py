"""Instruction mix per kernel of a gfx950 assembly file (hipcc -save-temps): counts by unit and the commonest opcodes."""
import collections
import re
import sys

text = open(sys.argv[1]).read()
only = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", text, re.S):
    name, body = m.group(1), m.group(2)
    if only and only not in name:
        continue
    ops = collections.Counter(l.split()[0] for l in body.split("\n")
                              if l.startswith("\t") and not l.strip().startswith((".", ";")))
    tot = lambda pre: sum(c for o, c in ops.items() if o.startswith(pre))
    print(name[:90])
    print("   valu", tot("v_"), "salu", tot("s_"), "ds", tot("ds_"), "vmem", tot(("global_", "buffer_", "flat_")))
    print("   ", ops.most_common(28))
