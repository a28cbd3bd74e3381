"""History matrices built on the device -- the counterpart of recbole_cdr/data/dataset.py:181-249 (get_history_matrix) as
CrossDomainDataset.history_user_matrix (:596-624, one row per ITEM) and history_item_matrix (:626-654, one row per USER) call it:
row r lists the ids r interacted with IN ORDER OF APPEARANCE in inter_feat, 0-padded to the longest row; sized by the union id
space.  The reference fills it with two Python loops over the interactions; here it is one stable device sort of the row ids
(positions inside a row = rank inside the sorted segment) and one scatter -- integer work, bit-identical to the reference
(golden-pinned through NATR's fixtures, tests/test_gpu_parity.py)."""
import torch


def history_matrix(user_ids, item_ids, n_users, n_items, row, device):
    """-> (history_matrix int64 [rows, width], history_value fp32 (1 where filled), history_len int64 [rows])."""
    user_ids = torch.as_tensor(user_ids, dtype=torch.int64).to(device)
    item_ids = torch.as_tensor(item_ids, dtype=torch.int64).to(device)
    if row == 'user':
        row_num, row_ids, col_ids = n_users, user_ids, item_ids
    else:
        row_num, row_ids, col_ids = n_items, item_ids, user_ids
    lens = torch.bincount(row_ids, minlength=row_num)
    width = int(lens.max()) if row_ids.numel() else 0
    mat = torch.zeros(row_num, width, dtype=torch.int64, device=device)
    val = torch.zeros(row_num, width, dtype=torch.float32, device=device)
    if row_ids.numel():
        order = torch.sort(row_ids, stable=True).indices
        srow = row_ids[order]
        start = torch.cumsum(lens, 0) - lens
        pos = torch.arange(srow.numel(), device=device) - start[srow]
        mat[srow, pos] = col_ids[order]
        val[srow, pos] = 1.0
    return mat, val, lens
