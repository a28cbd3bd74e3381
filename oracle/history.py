"""History matrices restated (oracle; test infrastructure only).  /root/reference recbole_cdr/data/dataset.py:
get_history_matrix :181-249 (rows in order of appearance in inter_feat, 0-padded, width = the longest row), called by
CrossDomainDataset.history_user_matrix :596-624 (row = item) and history_item_matrix :626-654 (row = user) with the UNION sizes.
value_field is None on every call site of the hot path (natr.py:77,87, dcdcsr.py:92-96, dtcdr.py:72-84) -> values are 1."""
import numpy as np
import torch


def history_matrix(user_ids, item_ids, n_users, n_items, row):
    user_ids, item_ids = np.asarray(user_ids), np.asarray(item_ids)
    if row == 'user':
        row_num, row_ids, col_ids = n_users, user_ids, item_ids
    else:
        row_num, row_ids, col_ids = n_items, item_ids, user_ids
    lens = np.zeros(row_num, dtype=np.int64)
    for r in row_ids:
        lens[r] += 1
    width = int(lens.max()) if len(lens) else 0
    mat = np.zeros((row_num, width), dtype=np.int64)
    val = np.zeros((row_num, width))
    lens[:] = 0
    for r, c in zip(row_ids, col_ids):
        mat[r, lens[r]] = c
        val[r, lens[r]] = 1.0
        lens[r] += 1
    return torch.LongTensor(mat), torch.FloatTensor(val), torch.LongTensor(lens)
