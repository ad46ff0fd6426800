"""CPU: the report files of the harness restatement.  CM_ is the reference's own table
(recognizer_test.py:491-499): a crosstab of the played names against themselves, in which every miss zeroes its
diagonal cell and marks the column of the name that came back."""
import csv
import os


def test_cm_table_follows_the_reference_rule(tmp_path):
    import pandas as pd
    from shazam_amd import harness
    played = ["a", "b", "c", "a", "b", "d"]
    result = ["a", "c", "c", "zzz", "b", "a"]        # misses: b -> c, a -> zzz (a name never played), d -> a
    rows = [{"file_name_played": p, "file_name_result": r} for p, r in zip(played, result)]
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        harness.write_reports(rows, "t.csv")
        # the rule stated with pandas the way the reference states it
        y_true, y_pred = pd.Series(played, name="Actual"), pd.Series(result)
        want = pd.crosstab(y_true, y_true).astype(object)
        for i in range(len(y_true)):
            if y_true[i] != y_pred[i]:
                want.at[y_true[i], y_true[i]] = 0
                want.at[y_true[i], y_pred[i]] = 1
        want.to_csv("want.csv")
        got_rows = list(csv.reader(open("CM_t.csv")))
        want_rows = list(csv.reader(open("want.csv")))
        as_num = lambda rs: [[c if i == 0 or j == 0 else (float(c) if c else None) for j, c in enumerate(r)] for i, r in enumerate(rs)]
        assert as_num(got_rows) == as_num(want_rows)
        for name in ("CMSK_t.csv", "CRSK_t.csv", "ASSK_t.csv"):
            assert os.path.getsize(name) > 0
    finally:
        os.chdir(cwd)
