#!/usr/bin/env python3
"""Copy the texture maps the reference checkout holds for head.obj into assets/ next to head.ptmesh (data files, byte for byte), so that the
GPU box -- which never sees /root/reference -- renders BASELINE configs[2] with the reference's own floor and wall maps.

head/head.mtl names six maps; three are in the checkout (concrete1.png = floor map_Kd, concrete1_normal.png = floor map_bump,
concrete2_normal.png = wall map_bump), three are missing large blobs (lambertian.png, normal.png, concrete2.png: SURVEY.md section 0)
and stay procedural stand-ins (pathtracer_amd/scene.py head_textures).  Run in the build container:
    python tools/import_head_textures.py [/root/reference/head]
"""
import hashlib
import os
import shutil
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/head"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")
NAMES = ["concrete1.png", "concrete1_normal.png", "concrete2_normal.png"]

os.makedirs(DST, exist_ok=True)
lines = []
for n in NAMES:
    shutil.copyfile(os.path.join(SRC, n), os.path.join(DST, n))
    os.chmod(os.path.join(DST, n), 0o644)
    lines.append("%s  %s" % (hashlib.sha256(open(os.path.join(DST, n), "rb").read()).hexdigest(), n))
open(os.path.join(DST, "SHA256SUMS"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
