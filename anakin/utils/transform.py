from artiboost_amd.models import batch_uvd2xyz, ortho6d_to_rotmat as compute_rotation_matrix_from_ortho6d  # noqa: F401  (anakin/utils/transform.py:512,578)
