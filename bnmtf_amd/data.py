"""Loaders on either side of the hot path: the drug-sensitivity text format of
data_drug_sensitivity/gdsc/load_data.py (cell line, cancer type, tissue, then one value per drug; an empty field is a
missing entry) with the per-entry Python loops vectorised, and the toy-data text matrices (numpy.loadtxt)."""
import numpy as np


def load_gdsc(location, sep=","):
    """load_data.py:15-54.  Returns (X, X_min, M, drug_names, cell_lines, cancer_types, tissues): X has 0 at the missing
    entries, X_min = X - (min(X) - 1) on the observed entries and 0 elsewhere, M the 0/1 mask."""
    lines = [line.split("\n")[0].split("\r")[0].split(sep) for line in open(location, 'r').readlines()]
    drug_names = lines[0][3:]
    cell_lines = [l[0] for l in lines[1:]]
    cancer_types = [l[1] for l in lines[1:]]
    tissues = [l[2] for l in lines[1:]]
    fields = np.array([l[3:] for l in lines[1:]], dtype=object)
    M = (fields != '').astype(float)
    X = np.zeros(M.shape)
    X[M == 1] = np.array(fields[M == 1], dtype=float)
    minimum = X.min() - 1
    X_min = np.where(M == 1, X - minimum, 0.0)
    return (X, X_min, M, drug_names, cell_lines, cancer_types, tissues)


def negate_gdsc(X, M):
    """load_data.py:57-70: negate and shift so that the smallest observed value of -X maps to 0."""
    Xn = -np.asarray(X, dtype=float)
    minimum = Xn.min()
    return np.where(np.asarray(M) != 0, Xn - minimum, 0.0)


def store_gdsc(location, X, M, drug_names, cell_lines, cancer_types, tissues):
    """load_data.py:73-86 (tab separated; nothing is written for a missing value)."""
    with open(location, 'w') as fout:
        fout.write("Cell Line\tCancer Type\tTissue\t" + "\t".join(drug_names) + "\n")
        for i, (cell_line, cancer_type, tissue, row) in enumerate(zip(cell_lines, cancer_types, tissues, X)):
            data = [str(val) if M[i][j] else "" for (j, val) in enumerate(row)]
            fout.write(cell_line + "\t" + cancer_type + "\t" + tissue + "\t" + "\t".join(data) + "\n")


def load_kernels(folder, file_names):
    """load_data.py:89-99: tab-separated square matrices with one header line."""
    return [np.array([line.split("\t") for line in open(folder + name, 'r').readlines()[1:]], dtype=float) for name in file_names]


def load_toy(folder):
    """R.txt / M.txt of a data_toy/ directory (generate_bnmf.py / generate_bnmtf.py write them with numpy.savetxt)."""
    return np.loadtxt(folder + "/R.txt"), np.loadtxt(folder + "/M.txt")
