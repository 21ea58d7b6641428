#!/usr/bin/env python3
"""Writes tests/golden/kat.json: the known-answer vectors the reference's own
files hold for the decoder path.  Data only -- numbers
quoted from READMEs / doc images / array literals, with their source lines:

  TS33C/README.md:110-111,296-303   rtl_433 pulses/gaps (samples @250 kHz), bitbuffer, packet, fields
  TS33C/docs/Decoding Description 2.png  90 raw bits -> packet -> +8.2 C / 53 %
  TFA/README.md:31                  tfrec line
  TFA/docs/Decoding_Description.png 88 raw bits -> 11 bytes
  somfy/decoder.py:18-27, somfy/README.md:30-35,50-56
  mumbi/transmitter.py:18-29, berner/README.md:45
  kwmobile/README.md:50-71          rtl_433 row + sniff.py line (Nexus PPM sensor)
Run from the repo root in the build container (reads nothing at run time).
"""
import json, os

kat = {
 "ts33c_readme": {
  "sample_rate_hz": 250000,
  "pulses": [241,231,226,102,105,104,226,106,99,104,110,104,103,103,105,228,230,105,104,106,103,103,226,225,107,226,108,106,227,106,233,107,109,104,106,229,104,108,109,108,226,226,104,107,228,104,108,109,229,104,229,231,230,229,227,105,107,226,102,227,104,105,107,104,109,107,105],
  "gaps": [257,261,143,140,261,140,139,145,139,137,259,143,141,138,140,258,138,139,141,139,265,139,264,260,139,138,258,141,259,134,136,137,139,260,137,140,138,134,137,260,264,140,137,261,260,138,135,137,261,139,257,258,259,262,138,138,262,142,263,139,138,139,261,136,136,138,2501],
  "bitbuffer_hex": "06bde3d0a975e18e40193bc0", "nbits": 90,
  "packet_hex": "9f210e5e16c239fb6711",
  "rolling_code": 1, "channel": 1, "battery_ok": True, "temperature": 21.6, "humidity": 39
 },
 "ts33c_png2": {
  "raw_groups": ["000001101","011110111","100011110","100001001","101111101","111111001","001101011","001100001","001101110","000111101"],
  "packet_hex": "9f210ede82c053f31387", "temperature": 8.2, "humidity": 53
 },
 "tfa_readme": {"packet_hex": "2dd471c98614366000565e", "identifier": "71C9", "temperature": 21.4, "humidity": 54, "battery_ok": True},
 "tfa_png": {
  "raw_bits": "1011010000101011100011101001001101100001100100001001110000000110000000000110101001111010",
  "packet_hex": "2dd471c98609396000565e", "identifier": "71C9", "temperature": 20.9, "humidity": 57, "battery_ok": True
 },
 "somfy_stop": {
  "manchester": [0,1,1,0,0,1,1,0,1,0,0,1,1,0,0,1,0,1,1,0,0,1,0,1,0,1,1,0,1,0,1,0,0,1,1,0,0,1,0,1,0,1,1,0,1,0,1,0,0,1,1,0,0,1,0,1,0,1,0,1,1,0,0,1,0,1,0,1,0,1,0,1,0,1,0,1,1,0,0,1,0,1,1,0,0,1,1,0,0,1,0,1,0,1,0,1,0,1,1,0,1,0,0,1,0,1,1,0,1,0,0,1],
  "payload_hex": "a5b8b8bdfdaf99", "frame_hex": "a51d0005405236", "checksum_ok": True,
  "control": 1, "rolling_code": 5, "address": 0x365240, "key": 0xA5
 },
 "somfy_readme2": {"frame_hex": "a3480003405236", "payload_hex": "a3ebebe8a8facc"},
 "somfy_readme_cksum": {"frame_hex": "a6161146bd5f36", "checksum_ok": True},
 "mumbi_codes": {"a_on":[0x19,0x10,0x0F,0x09,0x00],"a_off":[0x19,0x10,0x0E,0x08,0x00],"b_on":[0x19,0x10,0x0D,0x0A,0x00],"b_off":[0x19,0x10,0x0C,0x0B,0x00],"c_on":[0x19,0x10,0x0B,0x0E,0x00],"c_off":[0x19,0x10,0x0A,0x0F,0x00],"d_on":[0x19,0x10,0x07,0x01,0x00],"d_off":[0x19,0x10,0x06,0x00,0x00],"all_on":[0x19,0x10,0x04,0x03,0x00],"all_off":[0x19,0x10,0x08,0x0D,0x00]},
 "berner_example": {"code_hex": "02eaabbf"},
 "kwmobile_sniff": {"frame_hex": "808126f5e0", "bits": "100000001000000100100110111101011110",
                    "sensor_id": 128, "channel": 0, "battery_ok": True, "temperature": 29.4, "humidity": 94},
 "kwmobile_rtl433": {"frame_hex": "808111f600", "bits": "100000001000000100010001111101100000",
                     "sensor_id": 128, "channel": 0, "battery_ok": True, "temperature": 27.3, "humidity": 96}
}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json")
json.dump(kat, open(out, "w"), indent=1)
print("wrote", out)
