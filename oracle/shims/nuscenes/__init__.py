"""Import-only stand-in for the nuScenes devkit (absent from this image).  The dataset module of the reference
imports `get_prediction_challenge_split` to list scene tokens; the golden generator points it at its own
fabricated token list through TRAJSDE_FAKE_NUSCENES_SPLITS (json: {split: [tokens]})."""
