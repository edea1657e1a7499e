"""The WebSocket front end (diart_amd/ws.py) end to end over real sockets, with a recording engine
in place of the GPU: the reference's message format (base64 float32 text in, RTTM text out:
sources.py:204-271, utils.py:56-66, console/serve.py:119-124), several clients at once, control
frames, fragmentation, and clean-up when a client leaves."""
import base64
import os
import socket
import struct
import time

import numpy as np
import pytest

from diart_amd.serve import StreamServer
from diart_amd.ws import WebSocketFrontEnd, accept_key, decode_audio, encode_frame


class Engine:
    """One turn [start + 4.5, start + 5.0) for 'speaker' = slot per window whose newest sample > 0."""

    def reset(self, slot):
        pass

    def __call__(self, windows, starts, slots):
        return [np.array([[s + 4.5, s + 5.0, float(slot)]]) if w[-1] > 0 else np.zeros((0, 3))
                for w, s, slot in zip(windows, starts, slots)]


class Client:
    def __init__(self, port, path):
        self.s = socket.create_connection(("127.0.0.1", port), timeout=10)
        key = base64.b64encode(os.urandom(16)).decode()
        self.s.sendall((f"GET /{path} HTTP/1.1\r\nHost: x\r\nUpgrade: websocket\r\nConnection: Upgrade\r\n"
                        f"Sec-WebSocket-Key: {key}\r\nSec-WebSocket-Version: 13\r\n\r\n").encode())
        head = b""
        while b"\r\n\r\n" not in head:
            head += self.s.recv(1)
        assert b"101" in head.split(b"\r\n")[0] and accept_key(key).encode() in head
        self.buf = b""

    def send(self, opcode, payload, fin=True):
        frame = bytearray(encode_frame(opcode, payload, mask=os.urandom(4)))
        if not fin:
            frame[0] &= 0x7F
        self.s.sendall(bytes(frame))

    def send_audio(self, x):
        self.send(0x1, base64.b64encode(np.asarray(x, np.float32).tobytes()))

    def _need(self, n):
        while len(self.buf) < n:
            chunk = self.s.recv(65536)
            if not chunk:
                raise EOFError
            self.buf += chunk

    def recv(self):
        self._need(2)
        op, n = self.buf[0] & 0x0F, self.buf[1] & 0x7F
        assert not self.buf[1] & 0x80, "server frames are not masked"
        off = 2
        if n == 126:
            self._need(4)
            n, off = struct.unpack("!H", self.buf[2:4])[0], 4
        elif n == 127:
            self._need(10)
            n, off = struct.unpack("!Q", self.buf[2:10])[0], 10
        self._need(off + n)
        data, self.buf = self.buf[off:off + n], self.buf[off + n:]
        return op, data


def ramp(n, sign=1.0):
    return sign * (np.arange(n, dtype=np.float32) + 1) / 1e6


@pytest.fixture
def front():
    srv = StreamServer(None, None, max_streams=4, engine=Engine())
    fe = WebSocketFrontEnd(srv, port=0).start()
    yield fe, srv
    fe.stop()


def test_helpers():
    assert accept_key("dGhlIHNhbXBsZSBub25jZQ==") == "s3pPLMBiTxaQ9kYGzzhZRbK+xOo="      # RFC 6455 §1.3
    x = np.arange(5, dtype=np.float32)
    assert np.array_equal(decode_audio(base64.b64encode(x.tobytes()).decode()), x)
    assert np.array_equal(decode_audio(x.tobytes()), x)
    with pytest.raises(ValueError):
        decode_audio(b"abc")


def test_two_clients_get_their_own_rttm(front):
    fe, srv = front
    a, b = Client(fe.port, "alice"), Client(fe.port, "bob")
    for _ in range(200):
        if set(srv.open_streams) == {"alice", "bob"}:
            break
        time.sleep(0.01)
    assert set(srv.open_streams) == {"alice", "bob"}
    # alice: 6.0 s of positive audio in odd blocks -> windows at 0.0, 0.5, 1.0 s, one turn each;
    # bob: negative audio -> windows but no turns -> nothing is sent to him
    audio = ramp(96000)
    pos = 0
    for n in (12345, 30000, 40000, 13655):
        a.send_audio(audio[pos:pos + n])
        pos += n
    b.send_audio(ramp(96000, -1.0))
    a.send(0x9, b"hello")                                    # ping -> pong, in order with the data
    got, pong = [], False
    deadline = time.time() + 10
    while (len(got) < 3 or not pong) and time.time() < deadline:
        op, data = a.recv()
        if op == 0xA:
            pong = data == b"hello"
        else:
            assert op == 0x1
            got.append(data.decode())
    assert pong and len(got) == 3
    for i, line in enumerate(got):
        f = line.split()
        assert f[0] == "SPEAKER" and f[1] == "alice" and abs(float(f[3]) - (4.5 + 0.5 * i)) < 1e-6 and abs(float(f[4]) - 0.5) < 1e-6
    # a fragmented binary message (raw float32): one more block -> one more window
    raw = ramp(8000).astype("<f4").tobytes()
    a.send(0x2, raw[:10000], fin=False)
    a.send(0x0, raw[10000:], fin=True)
    op, data = a.recv()
    assert op == 0x1 and abs(float(data.decode().split()[3]) - 6.0) < 1e-6
    # closing handshake: the stream goes away, bob's stays
    a.send(0x8, struct.pack("!H", 1000))
    op, data = a.recv()
    assert op == 0x8
    for _ in range(200):
        if srv.open_streams == ["bob"]:
            break
        time.sleep(0.01)
    assert srv.open_streams == ["bob"]
    b.s.close()                                              # abrupt disconnect closes the stream too
    for _ in range(200):
        if not srv.open_streams:
            break
        time.sleep(0.01)
    assert srv.open_streams == [] and not fe.errors


def test_a_duplicate_stream_id_is_refused_without_touching_the_first(front):
    fe, srv = front
    a = Client(fe.port, "room")
    for _ in range(200):
        if srv.open_streams:
            break
        time.sleep(0.01)
    b = Client(fe.port, "room")                              # upgrade succeeds, then the server drops it
    with pytest.raises((EOFError, ConnectionError, socket.timeout, OSError)):
        b.s.settimeout(5)
        while True:
            b.recv()
    assert srv.open_streams == ["room"]
    a.send_audio(ramp(88000))
    op, data = a.recv()
    assert op == 0x1 and data.decode().split()[1] == "room"
    assert any("already open" in msg for _, msg in fe.errors)


def _close_code(client):
    client.s.settimeout(5)
    while True:
        op, data = client.recv()
        if op == 0x8:
            return struct.unpack("!H", data[:2])[0]


def test_protocol_violations_get_a_close_frame(front):
    """RSV bits, fragmented / oversized control frames are rejected with a close
    frame (1002) instead of a silent disconnect; the error list is bounded."""
    fe, srv = front
    c = Client(fe.port, "rsv")
    frame = bytearray(encode_frame(0x1, b"abcd", mask=os.urandom(4)))
    frame[0] |= 0x40                                          # RSV1 without a negotiated extension
    c.s.sendall(bytes(frame))
    assert _close_code(c) == 1002
    c = Client(fe.port, "ping")
    c.send(0x9, b"x", fin=False)                              # a control frame must not be fragmented
    assert _close_code(c) == 1002
    c = Client(fe.port, "bigping")
    c.send(0x9, b"y" * 126)                                   # nor longer than 125 bytes
    assert _close_code(c) == 1002
    assert fe.errors.maxlen is not None and len(fe.errors) == 3
    # an unsupported protocol version is refused during the upgrade
    s = socket.create_connection(("127.0.0.1", fe.port), timeout=5)
    s.sendall(b"GET /v HTTP/1.1\r\nHost: x\r\nUpgrade: websocket\r\nConnection: Upgrade\r\n"
              b"Sec-WebSocket-Key: AAAAAAAAAAAAAAAAAAAAAA==\r\nSec-WebSocket-Version: 8\r\n\r\n")
    assert b"426" in s.recv(200)


def test_backlog_is_bounded_by_back_pressure_then_policy_close():
    """A client far ahead of real time with a worker that never catches up: the connection is not
    read from beyond the backlog cap and is closed with 1008 after the timeout."""
    class Stuck(Engine):
        def __call__(self, windows, starts, slots):
            time.sleep(0.2)
            return super().__call__(windows, starts, slots)

    srv = StreamServer(None, None, max_streams=2, engine=Stuck())
    fe = WebSocketFrontEnd(srv, port=0, max_backlog_seconds=2.0, backlog_timeout=0.3).start()
    try:
        c = Client(fe.port, "flood")
        c.send_audio(ramp(16000 * 30))                        # 30 s at once: 51 windows >> the cap of 4
        assert _close_code(c) == 1008
        for _ in range(200):
            if not srv.open_streams:
                break
            time.sleep(0.01)
        assert srv.open_streams == []
    finally:
        fe.stop()
