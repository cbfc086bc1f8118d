"""Re-point the rarely used dense tiles of the ONE-SCENE shapes (12 view-instances per GPU) of the tracked tile table
at the popular ones, so that a step runs fewer distinct kernel symbols (instruction-cache footprint: profiles/
r04_experiments.txt #8).  Shapes that only the batched leg (24 / 48 instances) launches keep their measured choice.
    python tools/consolidate_table.py dualdiff_amd/tuned/gfx950.json out.json"""
import ast
import json
import sys

REMAP = {13: 52, 18: 52, 15: 52, 14: 52, 16: 20, 27: 28, 46: 44, 11: 12}
NO_GEGLU = {52}                                            # tile 52 has no GEGLU instantiation
ONE_SCENE_ROWS = {12, 40, 240, 336, 924, 1092, 1176, 4200}
L0_SHAPES = {(320, 320), (320, 640), (320, 960), (320, 1600), (960, 320), (1280, 320)}      # 16800 rows at 12 instances


def one_scene(key):
    return key[1] in ONE_SCENE_ROWS or (key[1] == 16800 and (key[2], key[3]) in L0_SHAPES)


def main(src, dst):
    table = json.load(open(src))
    n = 0
    for ent in table["entries"]:
        key = ast.literal_eval(ent[0])
        tile = ent[1][0]
        if key[0] != "g" or tile not in REMAP or not one_scene(key):
            continue
        if key[4] == 1 and REMAP[tile] in NO_GEGLU:
            continue
        ent[1] = [REMAP[tile]] + list(ent[1][1:])
        n += 1
    json.dump(table, open(dst, "w"), indent=0)
    print("re-pointed %d entries" % n)


if __name__ == "__main__":
    main(*sys.argv[1:3])
