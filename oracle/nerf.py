"""Oracle restatement of the reference's NeRF backbone builder (angles -> N, CA, C, O coordinates).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows structure_model/create_pdb.py:40-234 of the
reference (numpy path): float32 angles enter numpy trig as float32, the frame algebra runs in
float64 from the float64 initial coordinates -- kept as is.  Pinned by tests/golden/nerf.pt, generated
from the reference's own (ast-extracted) NERFBuilder / place_dihedral.
"""
import numpy as np

N_CA_LENGTH, CA_C_LENGTH, C_N_LENGTH, C_O_LENGTH = 1.46, 1.54, 1.34, 1.22      # create_pdb.py:23-26
N_INIT = np.array([17.047, 14.099, 3.625])                                        # create_pdb.py:29-32
CA_INIT = np.array([16.967, 12.784, 4.338])
C_INIT = np.array([15.685, 12.755, 5.133])
COLS = ["phi", "psi", "omega", "dihedral_o", "tau", "CA:C:1N", "1C:N:CA", "CA:C:O"]   # create_pdb.py:38


def place_dihedral(a, b, c, bond_angle, bond_length, torsion_angle):
    """create_pdb.py:175-234 (numpy branch): d such that |cd| = bond_length, angle(b,c,d) = bond_angle,
    dihedral(a,b,c,d) = torsion_angle."""
    unit = lambda x: x / np.linalg.norm(x, axis=-1)  # noqa: E731
    ab = b - a
    bc = unit(c - b)
    n = unit(np.cross(ab, bc))
    nbc = np.cross(n, bc)
    m = np.stack([bc, nbc, n], axis=-1)
    d = np.stack([-bond_length * np.cos(bond_angle),
                  bond_length * np.cos(torsion_angle) * np.sin(bond_angle),
                  bond_length * np.sin(torsion_angle) * np.sin(bond_angle)], axis=a.ndim - 1)
    return m.dot(d) + c


def backbone_coords(angles, center=True):
    """angles [l, 8] float32 in COLS order -> [4l, 3] float64 (N, CA, C, O per residue).

    = create_new_chain_nerf's NERFBuilder call (create_pdb.py:340-375): the three bond lengths stay
    at their constants, the four bond angles come from columns 4-7 (tau -> CA-C, CA:C:1N -> C-N,
    1C:N:CA -> N-CA, CA:C:O -> C-O), then NERFBuilder.cartesian_coords (create_pdb.py:104-155)."""
    phi, psi, omega, o_dih = (angles[:, i] for i in range(4))
    ang_ca_c, ang_c_n, ang_n_ca, ang_c_o = (angles[:, i] for i in range(4, 8))
    bb = [N_INIT.copy(), CA_INIT.copy(), C_INIT.copy()]
    dih = np.stack([psi[:-1], omega[:-1], phi[1:]]).T
    bonds = ((C_N_LENGTH, ang_c_n), (N_CA_LENGTH, ang_n_ca), (CA_C_LENGTH, ang_ca_c))
    for i in range(dih.shape[0]):
        for j, (length, ang) in enumerate(bonds):
            bb.append(place_dihedral(bb[-3], bb[-2], bb[-1], ang[i], length, dih[i][j]))
    out = []
    for i in range(0, len(bb), 3):
        n, ca, c = bb[i:i + 3]
        out.extend([n, ca, c, place_dihedral(n, ca, c, ang_c_o[i // 3], C_O_LENGTH, o_dih[i // 3])])
    coords = np.array(out)
    return coords - coords.mean(axis=0) if center else coords
