#!/usr/bin/env python3
"""Convert the reference's head/head.obj scene asset into assets/head.ptmesh.

The GPU box never sees /root/reference, so the scene every BASELINE.json config
is quoted on travels as a compact binary *data* file (no reference source).
Run in the authoring container only:  python tools/convert_head_obj.py

.ptmesh layout (little endian):
  char[8]  magic "PTMESH1\\0"
  int32    numVertices, numTriangles, numMaterials, reserved
  float32  position[numVertices][3]
  float32  normal  [numVertices][3]
  float32  uv      [numVertices][2]
  int32    index   [numTriangles][3]
  int32    triMaterial[numTriangles]
  material[numMaterials]: char name[32]; float diffuse[4]; float specular[3];
                          float glossiness; char map_kd[64]; char map_bump[64]

Import conventions (the reference's FW importer is not in its tree, so these
are this build's definitions, SURVEY.md 8c "parity unpinned"): one vertex per
distinct v/vt/vn triple in order of first use, triangles in file order (which is
submesh order for head.obj: head, wall, floor), vt.y not flipped, material
defaults Kd=(.75,.75,.75,1) Ks=(0,0,0) Ns=0.
"""
import os
import struct
import sys

import numpy as np

SRC = "/root/reference/head/head.obj"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets", "head.ptmesh")


def parse_mtl(path):
    mats, cur = [], None
    for line in open(path):
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "newmtl":
            cur = dict(name=tok[1], diffuse=[0.75, 0.75, 0.75, 1.0], specular=[0.0, 0.0, 0.0],
                       gloss=0.0, map_kd="", map_bump="")
            mats.append(cur)
        elif tok[0] == "Kd":
            cur["diffuse"][:3] = [float(x) for x in tok[1:4]]
        elif tok[0] == "Ks":
            cur["specular"] = [float(x) for x in tok[1:4]]
        elif tok[0] == "Ns":
            cur["gloss"] = float(tok[1])
        elif tok[0] == "map_Kd":
            cur["map_kd"] = tok[1]
        elif tok[0] in ("map_bump", "bump"):
            cur["map_bump"] = tok[1]
    return mats


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else SRC
    dst = sys.argv[2] if len(sys.argv) > 2 else DST
    v, vt, vn = [], [], []
    verts, vert_index, tris, tri_mat = [], {}, [], []
    mats, mat_id = [], 0
    for line in open(src):
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "v":
            v.append(tuple(float(x) for x in tok[1:4]))
        elif tok[0] == "vt":
            vt.append(tuple(float(x) for x in tok[1:3]))
        elif tok[0] == "vn":
            vn.append(tuple(float(x) for x in tok[1:4]))
        elif tok[0] == "mtllib":
            mats = parse_mtl(os.path.join(os.path.dirname(src), tok[1]))
        elif tok[0] == "usemtl":
            mat_id = [m["name"] for m in mats].index(tok[1])
        elif tok[0] == "f":
            assert len(tok) == 4, "asset is already triangulated"
            tri = []
            for corner in tok[1:]:
                key = tuple(int(x) for x in corner.split("/"))
                if key not in vert_index:
                    vert_index[key] = len(verts)
                    verts.append(key)
                tri.append(vert_index[key])
            tris.append(tri)
            tri_mat.append(mat_id)
    pos = np.array([v[k[0] - 1] for k in verts], dtype=np.float32)
    uv = np.array([vt[k[1] - 1] for k in verts], dtype=np.float32)
    nrm = np.array([vn[k[2] - 1] for k in verts], dtype=np.float32)
    idx = np.array(tris, dtype=np.int32)
    tm = np.array(tri_mat, dtype=np.int32)
    with open(dst, "wb") as f:
        f.write(b"PTMESH1\0")
        f.write(struct.pack("<4i", len(verts), len(tris), len(mats), 0))
        f.write(pos.tobytes()); f.write(nrm.tobytes()); f.write(uv.tobytes())
        f.write(idx.tobytes()); f.write(tm.tobytes())
        for m in mats:
            f.write(struct.pack("<32s4f3ff64s64s", m["name"].encode(), *m["diffuse"], *m["specular"],
                                m["gloss"], m["map_kd"].encode(), m["map_bump"].encode()))
    print("wrote %s: %d vertices, %d triangles, %d materials, %d bytes" %
          (dst, len(verts), len(tris), len(mats), os.path.getsize(dst)))


if __name__ == "__main__":
    main()
