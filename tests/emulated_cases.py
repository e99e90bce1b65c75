"""Driver of tests/test_library_emulated.py (run as a script in a process of its own, because the library
reads OVS_RESOLVE once per process): points openvslam_amd._lib at the library built for the host
(tools/emu) and calls the parity checks of the GPU test files with small inputs.  Prints the kernels that
were launched and `OK <n>`.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))

import host_lib  # noqa: E402

L = host_lib.install()

from openvslam_amd import _abi as A  # noqa: E402
from openvslam_amd import _lib  # noqa: E402
import test_area_matcher as TA  # noqa: E402
import test_fuse_matcher as TF  # noqa: E402
import test_gpu_extractor as TE  # noqa: E402
import test_gpu_hamming_knn as TK  # noqa: E402
import test_gpu_stereo as TS  # noqa: E402
import test_landmark_descriptor as TL  # noqa: E402
import test_gpu_bow as TB  # noqa: E402
import test_gpu_projection as TP  # noqa: E402
import test_gpu_robust_bow as TR  # noqa: E402


def matchers():
    yield lambda: TP.test_match_current_and_last_frames(1, 600, 800, A.DIR_NONE, False, False, False, 1)
    yield lambda: TP.test_match_current_and_last_frames(2, 500, 700, A.DIR_FORWARD, True, True, True, 1)
    yield lambda: TP.test_match_current_and_last_frames(3, 400, 600, A.DIR_BACKWARD, True, True, True, 0)
    yield lambda: TP.test_match_current_and_last_frames(5, 50, 400, A.DIR_NONE, False, True, False, 1)
    yield lambda: TP.test_match_current_and_last_frames(6, 0, 10, A.DIR_NONE, False, False, False, 1)
    yield lambda: TP.test_match_current_and_last_frames(7, 300, 0, A.DIR_NONE, False, False, False, 1)
    yield lambda: TP.test_match_frame_and_landmarks(11, 600, 1200, False, False)
    yield lambda: TP.test_match_frame_and_landmarks(13, 100, 1000, False, True)
    yield lambda: TP.test_match_frame_and_landmarks(12, 500, 1500, True, True)
    yield lambda: TP.test_match_frame_and_keyframe(21, 600, 700, 100, 1)
    yield lambda: TP.test_match_frame_and_keyframe(23, 500, 900, 64, 1)
    yield lambda: TP.test_match_by_sim3_transform(32, 500, 900)
    yield TP.test_greedy_conflicts_exhaust_candidate_lists
    yield TP.test_chain_of_forty_queries_that_push_each_other_on
    yield lambda: TR.test_brute_force_match(3, 64, 900, 0.75, 1)
    yield lambda: TR.test_brute_force_match(1, 500, 500, 0.75, 1)
    yield lambda: TR.test_bow_match_frame_and_keyframe(13, 300, 300, 10, 1)
    yield lambda: TR.test_bow_match_frame_and_keyframe(12, 500, 800, 60, 0)
    yield lambda: TR.test_bow_matchers_on_clustered_buckets(23, 500, 450, 8, 0)
    yield lambda: TR.test_bow_matchers_on_clustered_buckets(22, 400, 700, 20, 1)
    yield TR.test_bow_out_of_range_index
    yield lambda: TF.test_gpu_matches_oracle(1, 600, 900, True, 3.0)
    yield lambda: TF.test_gpu_matches_oracle(3, 40, 200, True, 10.0)
    yield lambda: TF.test_gpu_matches_oracle(4, 0, 5, False, 3.0)


def bow():
    yield lambda: TB.test_transform_matches_oracle(8, 3, 700)
    yield lambda: TB.test_transform_matches_oracle(10, 6, 500)
    yield lambda: TB.test_weighting_and_scoring_types(A.BOW_TF, A.BOW_L2_NORM)
    yield lambda: TB.test_irregular_trees_and_every_node_level(2)
    yield TB.test_wide_nodes_and_small_inputs


def extract():
    yield lambda: TE.test_extract_matches_oracle(640, 480, 1000)
    yield lambda: TE.test_blur_borders_every_width_class(259, 0)
    yield lambda: TE.test_blur_borders_every_width_class(514, 1)
    yield TE.test_extract_low_texture_uses_min_threshold
    yield TE.test_extract_with_rectangle_masks
    yield TE.test_extract_portrait
    yield TE.test_extract_from_strided_host_buffers
    yield lambda: TE.test_extract_large_quota_per_level(2, 2500, 640, 480)
    yield lambda: TS.test_stereo_compute_matches_oracle(640, 480, 800, 4)


def others():
    yield lambda: TK.test_hamming_pairs(4097)
    yield lambda: TK.test_hamming_matrix(257, 129)
    yield lambda: TK.test_get_keypoints_in_cell(1, 2000, 500)
    yield lambda: TK.test_desc_db_search(3, 4097, 65)
    yield lambda: TS.test_match_for_triangulation(2, 800, 20, 0)
    yield TS.test_match_for_triangulation_borderline_epipolar_decisions
    yield lambda: TA.test_gpu_contested_keypoints(2, 1500, 40, 0, 1.0, 200)
    yield TA.test_gpu_closer_keypoint_takes_over
    yield lambda: TL.test_gpu_matches_oracle(2, [300, 1, 1000, 2])


def main():
    group = sys.argv[1] if len(sys.argv) > 1 else "matchers"
    L.ovs_timing_reset()
    L.ovs_timing_enable(1)
    n = 0
    for case in {"matchers": matchers, "bow": bow, "extract": extract, "others": others}[group]():
        case()
        n += 1
    L.ovs_timing_enable(0)
    print("kernels:", " ".join("%s=%d" % (k, v[1]) for k, v in sorted(_lib.timing_report().items())))
    print("OK", n)


if __name__ == "__main__":
    main()
