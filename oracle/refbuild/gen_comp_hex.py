"""oracle/refbuild: writes the `<name>.comp.hex.h` headers that src/rife.cpp:9-21 and src/warp.cpp:5-7 include.  In the reference they are
generated at build time from the GLSL sources by src/generate_shader_comp_header.cmake (strip the leading comment, strip indentation, hex dump
into `static const char <name>_comp_data[]`); this does the same, from the .comp files where they lie.  The arrays are only ever passed to
compile_spirv_module, which the CPU path never reaches."""
import os
import re
import sys

src, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
for f in sorted(os.listdir(src)):
    if not f.endswith(".comp"):
        continue
    name = f[:-5]
    text = open(os.path.join(src, f)).read()
    text = text[text.find("#version"):]
    text = re.sub(r"\n +", "\n", text)
    body = ",".join("0x%02x" % b for b in text.encode())
    with open(os.path.join(out, name + ".comp.hex.h"), "w") as fh:
        fh.write("static const char %s_comp_data[] = {%s};\n" % (name, body))
