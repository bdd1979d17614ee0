#!/usr/bin/env python3
"""profiles/algorithmic_ops.json: the arithmetic ONE PATH of a bench configuration needs, to price the kernels against (bench.py: roofline.algorithmic).

    python tools/algorithmic_ops.py

Inputs (all under profiles/, produced by oracle/opcount.py on the CPU and tools/traversal_stats.py on the GPU):
  r03_oracle_opcount_<cfg>.json   executed arithmetic of the CPU oracle per path, exact (basic-block counts on its LLVM IR), split into `path` (sampler, camera
                                  ray, surface interaction, emitter sampling, BSDF, modulation, MIS, roulette) and `query` (the ray queries, BRUTE FORCE in the oracle)
  r03_traversal_stats_<cfg>.json  what the product's traversal does per ray (TLAS node steps, leaf visits, triangle tests), counted by the kernels themselves
One op = one arithmetic instruction (an fma is one op; loads, stores, address arithmetic and control flow are not counted).
  * scenes the product ALSO traces by testing every object (rectangle-only scenes of a few objects, trace_flat: C2 / C3): ops = path + query of the oracle;
  * scenes behind a BVH (C4 / C5): ops = path (oracle) + rays_per_path x [node steps x NODE + leaf visits x LEAF + triangle tests x TRI], with the per-unit
    costs below -- LEAF and TRI from the oracle's own per-call counts, NODE counted on box_entry / node_step (dtof_traverse.h).
"""
import json, os
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(HERE, "profiles")
NODE_STEP_OPS = 44        # two slab tests (6 multiply-adds + 6 min / max + 3 + 3 min / max + compare = 19 each, the form of round 4) + ordering / selection of the children (6)


def newest(pattern):      # the newest round's file that exists
    for tag in ("r06", "r05", "r04", "r03"):
        if os.path.exists(os.path.join(P, pattern % tag)): return pattern % tag
    return pattern % "r05"


def load(name):
    path = os.path.join(P, name)
    return json.load(open(path)) if os.path.exists(path) else None


out = {}
c2_name = newest("%s_oracle_opcount_c2.json"); c2 = load(c2_name)
if c2:
    for cfg in ("c2", "c3"):
        out[cfg] = {"ops_per_path": c2["ops_per_path_total"], "source": "profiles/%s (oracle sha %%s): path + query groups; the product tests every object too (trace_flat)" % c2_name % c2["oracle_sha16"],
                    "breakdown": {"path_logic": c2["groups_per_path_total"]["path"], "ray_queries_all_objects": c2["groups_per_path_total"]["query"], "by_category": c2["ops_per_path"]}}
c4_name, t4_name = newest("%s_oracle_opcount_c4.json"), newest("%s_traversal_stats_c4.json"); c4, t4 = load(c4_name), load(t4_name)
if c4 and t4:
    per_call = c4["ops_per_call"]
    leaf = per_call.get("instance_to_world", 85) + per_call.get("m_affine_inverse", 49) + per_call.get("m_point", 9) + per_call.get("m_vector", 9) + 26 + 3   # + the mesh's own slab test and the reciprocals
    tri = per_call.get("tri_intersect", 38)
    per_ray = t4["tlas_node_steps_per_ray"] * NODE_STEP_OPS + t4["leaf_visits_per_ray"] * leaf + t4["triangle_tests_per_ray"] * tri
    trav = t4["rays_per_path"] * per_ray
    for cfg in ("c4", "c5"):
        out[cfg] = {"ops_per_path": round(c4["groups_per_path_total"]["path"] + trav, 1),
                    "source": "profiles/%s (oracle sha %s, path group) + profiles/%s x per-unit costs (node step %d, leaf visit %.0f, triangle test %.0f ops)"
                              % (c4_name, c4["oracle_sha16"], t4_name, NODE_STEP_OPS, leaf, tri),
                    "breakdown": {"path_logic": c4["groups_per_path_total"]["path"], "traversal": round(trav, 1), "rays_per_path": round(t4["rays_per_path"], 3),
                                  "per_ray": {"tlas_node_steps": round(t4["tlas_node_steps_per_ray"], 2), "leaf_visits": round(t4["leaf_visits_per_ray"], 3),
                                              "triangle_tests": round(t4["triangle_tests_per_ray"], 2), "ops": round(per_ray, 1)},
                                  "oracle_brute_force_queries_for_comparison": c4["groups_per_path_total"]["query"]}}
json.dump(out, open(os.path.join(P, "algorithmic_ops.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
