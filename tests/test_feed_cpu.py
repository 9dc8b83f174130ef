"""CPU: bag assembly of the feed against the reference's recipe (datasets/dataset_survival.py:354-366)."""
import torch

from multimodalfusion_amd.feed import load_slide_bags


def test_multi_slide_bag_equals_torch_cat(tmp_path):
    torch.manual_seed(0)
    bags = [torch.randn(n, 1024) for n in (5, 1, 17)]
    paths = []
    for i, b in enumerate(bags):
        p = tmp_path / f"slide_{i}.pt"
        torch.save(b, p)
        paths.append(str(p))
    got = load_slide_bags(paths, pin=False)
    assert torch.equal(got, torch.cat(bags, dim=0))                 # what the reference's __getitem__ returns
    assert load_slide_bags(paths, pin=False, dtype=torch.bfloat16).dtype == torch.bfloat16
    assert torch.equal(load_slide_bags([], pin=False), torch.zeros((1, 1)))      # "pathology missing" sentinel


def test_bf16_on_disk_round_trip(tmp_path):
    b = torch.randn(9, 1024).to(torch.bfloat16)
    torch.save(b, tmp_path / "s.pt")
    got = load_slide_bags([str(tmp_path / "s.pt")], pin=False)
    assert got.dtype == torch.bfloat16 and torch.equal(got, b)
