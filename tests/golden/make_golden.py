"""Collects the reference's DATA fixtures for the NLS path into tests/golden/ and writes the
expected values the reference records for them.  Run once in the build container (the only
place /root/reference exists); the outputs are committed, this script documents their origin.

  st3_calib/1..9.txt     <- st3-calibration/calib/1..9.txt      chessboard corners (data files)
  st16_odom/odometryInfo.txt <- st16-pcl-viewer/data/odom_lidar/odometryInfo.txt   70 poses (data file)
  st7_ransac/*.csv       <- st7-ransac/data/{good,bad}.csv      parabola samples (data files)
  st6_icp/*.csv          <- st6-icp/log/binding/{pc1,pc2,pc1_prime_1,pc1_prime_2}.csv   inputs and recorded iterates (data files)
  known_answers.json     <- values the reference itself publishes:
       st7-ransac/pyDraw/drawerResult.py:12-16, st17-ceres/img/{release,debug}.png (transcribed
       in BASELINE.md), st17-ceres/src/ceres_bound.cpp:26-65, SURVEY.md section 4 (calibration
       numbers from the survey-time restatement of calib.cpp).
No reference SOURCE text is copied -- only data files and numbers.
"""
import json
import os
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    for i in range(1, 10):
        shutil.copy(f"{REF}/st3-calibration/calib/{i}.txt", f"{HERE}/st3_calib/{i}.txt")
    os.makedirs(f"{HERE}/st16_odom", exist_ok=True)
    shutil.copy(f"{REF}/st16-pcl-viewer/data/odom_lidar/odometryInfo.txt", f"{HERE}/st16_odom/odometryInfo.txt")
    for f in ("good.csv", "bad.csv"):
        shutil.copy(f"{REF}/st7-ransac/data/{f}", f"{HERE}/st7_ransac/{f}")
    os.makedirs(f"{HERE}/st6_icp", exist_ok=True)
    for f in ("pc1.csv", "pc2.csv", "pc1_prime_1.csv", "pc1_prime_2.csv"):
        shutil.copy(f"{REF}/st6-icp/log/binding/{f}", f"{HERE}/st6_icp/{f}")
    ka = {
        "st7_parabola": {
            "source": "st7-ransac/pyDraw/drawerResult.py:12-16",
            "leastSquare_good": [0.9645, 1.93589, 3.0065],
            "leastSquare_bad": [0.509129, 1.0617, 3.61135],
            "gaussNewton_good_10": [0.964502, 1.93589, 3.0065],
            "gaussNewton_bad_10": [0.509132, 1.06171, 3.61135],
        },
        "st17_pnp": {
            "source": "st17-ceres/img/release.png, img/debug.png, src/main.cpp:17-32",
            "q_true_xyzw": [0.40958, 0.70941, -0.49673, -0.28679],
            "t_true": [3.0, 2.0, 1.0],
            "q_init_xyzw": [0.45452, 0.54168, -0.54168, -0.45452],
            "t_init": [2.5, 0.0, 0.0],
            "release_iterations": {"DynamicAutoDiff": 6, "AutoDiff": 6, "SizedCostFunction": 8, "SelfGaussNewton": 7},
            "release_initial_cost": 2.232755,
            "final_cost_below": 1e-17,
        },
        "st17_ceres_bound": {"source": "st17-ceres/src/ceres_bound.cpp:26-65", "x_free": 3.0, "x_bounded": 2.0,
                             "lower": -2.0, "upper": 2.0, "x0": 0.0},
        "st3_calibration": {
            "source": "SURVEY.md section 4 (survey-time numpy restatement of st3-calibration/src/src/calib.cpp)",
            "board_square_m": 0.028,
            "init_fx_fy_u0_v0": [3061.6206, 3060.0818, 2010.2196, 1473.5033],
            "sse_first": 1736.8916, "sse_last": 133.5132, "gn_iterations": 8,
            "final_intr_dist": [3038.238, 3037.528, 2004.882, 1468.111, 0.208026, -1.393321, 2.492627, 1.634e-6, -9.5908e-4],
        },
    }
    with open(f"{HERE}/known_answers.json", "w") as f:
        json.dump(ka, f, indent=1)


if __name__ == "__main__":
    main()
