"""Writes this directory's fixture: the first FRAMES views of the reference's bundled scene
(/root/reference/data/r_<i>.png, byte for byte) and a transforms.json holding exactly those frames'
entries of /root/reference/data/transforms.json (same key order, same numbers).

load_dataset (reference data.c:216-257) pairs r_<i>.png with frames[i], so a prefix of the scene is
itself a valid scene.  Run in the build container, where the reference checkout is mounted:
    python tests/golden/scene/make_scene_fixture.py
"""
import json
import os
import shutil

FRAMES = 4
SRC = "/root/reference/data"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    meta = json.load(open(os.path.join(SRC, "transforms.json")))
    meta["frames"] = meta["frames"][:FRAMES]
    with open(os.path.join(HERE, "transforms.json"), "w") as f:
        json.dump(meta, f, indent=4)
        f.write("\n")
    for i in range(FRAMES):
        shutil.copyfile(os.path.join(SRC, f"r_{i}.png"), os.path.join(HERE, f"r_{i}.png"))
        os.chmod(os.path.join(HERE, f"r_{i}.png"), 0o644)


if __name__ == "__main__":
    main()
