#!/usr/bin/env python3
"""Build a variant of libgcre_hip.so for an A/B measurement on the GPU box:
    tools/build_variant.py NAME [extra hipcc flags...]   ->   geneticscre_amd/variants/libgcre_hip_NAME.so
    GCRE_LIB=geneticscre_amd/variants/libgcre_hip_NAME.so python bench.py ...
Variants are git-ignored (*.so) and travel with the gpurun snapshot like the product library."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geneticscre_amd import build as b  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(b.PKG, "variants", f"libgcre_hip_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
print(b.build(force=True, out=out, extra=extra))
